// Internal launch interface shared by the kernel translation units and the
// C-ABI layer.  Not installed; the public interface is include/pdekan.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pdekan {

// Kernels with more than 48 KB of dynamic LDS need the limit raised once per (device, kernel); later launches with
// no more than what was granted skip the runtime call.  (pdekan_api.hip)
void ensure_dynamic_lds(const void* kernel, size_t bytes);

// One KANLinear layer as the kernels see it (device pointers, caller-owned).
struct LayerView {
  int I, O, G, act;
  const float* grid;       // [I][G+7]
  const float* base_w;     // [O][I]
  const float* spline_w;   // [O][I][G+3]
  const float* scaler;     // [O][I] or nullptr
  const float* ginfo;      // [I]: per feature 1/h (knots equally spaced), -1/h (nearly) or 0, see knot_spacing_info; nullptr = unknown
  int C() const { return G + 3; }
  int C1() const { return G + 4; }
  int nk() const { return G + 7; }
  size_t packed_floats() const { return (size_t)O * I * C1(); }
};

// A launch that the 12-jet matrix kernels of kan_f16.hip replace when every knot row of the layer is (nearly) equally
// spaced (settled on the device from the layer's ginfo): with ginfo set, the kernel returns at once in that case.
struct JetSkip { const float* ginfo = nullptr; int I = 0; };

// Seeding of a model's input jets (launch_seed_jets) as a job the first layer's forward can carry: the kernel builds
// the jets of its inputs from the points x (value, unit first derivative in the feature's own direction) instead of
// reading them, and leaves them at a_out for the backward pass.
struct SeedSrc { const float* x = nullptr; int64_t sn = 0, sd = 0; float* a_out = nullptr; };
// the loss reduction of a step (launch_loss_finish) as a job another launch can carry
struct LossJob { const float* part = nullptr; int nparts = 0; float inv_n = 0.f; float* out = nullptr; int accumulate = 0; };
struct ChunkSelect { const float* ginfo = nullptr; int I = 0, first_uniform = 0, n_uniform = 0, first_other = 0, n_other = 0; };

// kan_kernels.hip
void launch_grid_info(const LayerView& L, float* ginfo, hipStream_t st);
// out[N][I][G+3] = order-th derivative of every cubic basis at x[N][I]
// (spline_order 0 .. 5; 3 takes the local cubic evaluation)
void launch_basis_dense(const float* x, const float* grid, float* out, int64_t N, int I, int G, int order, hipStream_t st,
                        int spline_order = 3);
// out4[e] = (value or 0, first k, first 5 in pass 0 else 0, second k) of in12[e], e < elems, k = 0..4
void launch_gather_jets4(const float* in12, float* out4, int64_t elems, int k, bool keep_value, hipStream_t st,
                         JetSkip skip = {});
// out12[e] <- pass k's (value share, first k, time in pass 0, second k); the value shares are summed over k = 0..4
void launch_scatter_jets4(const float* in4, float* out12, int64_t elems, int k, hipStream_t st, JetSkip skip = {});
// curve2coeff: coeff[O][I][G+3] from samples x[N][I] (element strides) and either y[N][I][O] (element strides) or,
// with y = nullptr, the edge functions of the layer (old_w, old_scaler or nullptr) on old_grid; false: grid too fine
bool launch_fit_coefficients(const float* x, int64_t xsn, int64_t xsi, int64_t N, int I, int O, int G, const float* grid,
                             const float* y, int64_t ysn, int64_t ysi, int64_t yso, const float* old_grid,
                             const float* old_w, const float* old_scaler, float* coeff, hipStream_t st);
void launch_pack_weights(const LayerView& L, float* wp, float* wt, hipStream_t st, float* ginfo_out = nullptr);
constexpr int kPackLayersMax = 8;
bool launch_pack_model(const LayerView* layers, int n_layers, float* const* wp, float* const* wt, float* const* ginfo,
                       unsigned* clear, int clear_words, hipStream_t st);
bool launch_seed_jets(int D, int D2, const float* x, int64_t stride_n, int64_t stride_d, float* a, int64_t N,
                      int I, hipStream_t st);
// amax_out: where the per-slot maxima of y (1 + D + D2 words) are combined into, or nullptr
bool launch_fwd_generic(int D, int D2, const LayerView& L, const float* a, const float* wt, float* y, int64_t N,
                        hipStream_t st, bool skip_near_uniform = false, unsigned* amax_out = nullptr,
                        SeedSrc seed = SeedSrc());
// whether the forward kernels above / below take a SeedSrc for this layer (else the caller seeds first)
bool fwd_generic_takes_seed(int D, int D2, const LayerView& L);
bool fwd_sparse_takes_seed(const LayerView& L);
bool launch_bwd_input_generic(int D, int D2, const LayerView& L, const float* a, const float* gy,
                              const float* wp, float* ga, int64_t N, hipStream_t st, bool skip_near_uniform = false);
int generic_weight_chunks(int64_t N, int I, int O);
// seeded_input: a holds the seed jets of a model's first layer (value, 1 in the input's own direction, zeros)
bool launch_bwd_weight_generic(int D, int D2, const LayerView& L, const float* a, const float* gy, float* part,
                               int nchunks, int64_t N, hipStream_t st, bool skip_near_uniform = false,
                               bool seeded_input = false);
// first layers of 12-jet models that have kernels of their own behind the two launchers above
bool seeded_first_layer(int D, int D2, const LayerView& L);
// sum_buf: packed_floats() floats of scratch, used when nchunks > 1 or with sel
// sel.ginfo given: the chunks summed are [first_uniform, + n_uniform) when every knot row of the layer is (nearly)
// equally spaced, else [first_other, + n_other) - settled on the device (nchunks is not used then)
void launch_reduce_unpack(const LayerView& L, const float* part, int nchunks, float* sum_buf, float* g_base,
                          float* g_spline, float* g_scaler, int accumulate, hipStream_t st, ChunkSelect sel = {},
                          LossJob loss = LossJob());
bool launch_bs_residual(int D, const float* x, int64_t stride_n, int64_t stride_d, const float* vjet,
                        const float* target, float sigma, float rate, float inv_n, float* res, float* gy,
                        float* loss_part, int nblocks, int64_t N, hipStream_t st);
void launch_loss_finish(const float* loss_part, int nparts, float inv_n, float* loss_out, int accumulate,
                        hipStream_t st);

// kan_head.hip: last layer (one output) + residual + loss + its backward, fused
constexpr int kHeadBlocksMax = 1280;     // most blocks head_blocks() returns: one loss partial and one gradient table each
bool head_eligible(int D, const LayerView& L);
int head_blocks(int64_t N, const LayerView& L);
void launch_head_bs(int D, const LayerView& L, const float* a, const float* x, int64_t stride_n, int64_t stride_d,
                    const float* target, const float* wp, float sigma, float rate, float inv_n, float* res,
                    float* ga, float* part, float* loss_part, int nblocks, int64_t N, hipStream_t st,
                    unsigned* cot_out = nullptr);   // per-jet maxima of ga (4 or 12 words) are combined into cot_out

// kan_small.hip: one Black-Scholes step of a small model (inputs (S, t), one output, layers no wider than 8) in two
// launches: thread = point through every layer and back, then one reduction / unpack launch
constexpr int kSmallLayersMax = 4;
struct SmallGrads { float *base_w, *spline_w, *scaler; };
bool small_step_eligible(const LayerView* layers, int nl, size_t* lds = nullptr);
size_t small_step_part_floats(const LayerView* layers, int nl, int64_t N);      // workspace: the blocks' partial sums
// grads = nullptr: loss / residual only; res_out, loss_out, target may be nullptr
void launch_small_step(const LayerView* layers, const SmallGrads* grads, int nl, const float* x, int64_t sn, int64_t sd,
                       const float* target, int64_t N, float sigma, float rate, float inv_n, float* res_out,
                       float* loss_out, float* part, int accumulate, hipStream_t st);

// kan_deeponet.hip: one training step of the physics-informed DeepONet of run1.py:75-149 (sensor networks model1 and
// model2, mixing network model4, trunk model5; all layers but model4's first no wider than 8) in four launches
constexpr int kDonLayersMax = 4;
struct DonNetView { const LayerView* layers; const SmallGrads* grads; int nl; };   // grads: one per layer, or nullptr
// nullptr when the four networks (model1, model2, model4, model5) and sizes fit the kernels, else the reason
const char* deeponet_step_refusal(const DonNetView nets[4], int64_t sensors, int64_t n_bc, int64_t n_pde);
size_t deeponet_step_ws_floats(const DonNetView nets[4], int64_t sensors, int64_t n_bc, int64_t n_pde);   // 0: refused
// losses[3] = (pde + bc_weight * bc, pde, bc); s_pred [n_bc], res_out [n_pde] and losses may be nullptr
bool launch_deeponet_step(const DonNetView nets[4], const float* u1, const float* u2, int64_t sensors, const float* x_bc,
                          const float* t_bc, const float* out_bc, int64_t n_bc, const float* x_pde, const float* t_pde,
                          const float* out_pde, int64_t n_pde, float sigma, float rate, float bc_weight, float* losses,
                          float* s_pred, float* res_out, float* ws, hipStream_t st);

// kan_sparse.hip: sparse-support VALU path for hidden layers (J = 4 jets): forward and input gradients
bool sparse_eligible(int D, int D2, const LayerView& L);
bool sparse_input_grad_eligible(int D, int D2, const LayerView& L);
size_t sparse_table_floats(const LayerView& L);
void launch_pack_window(const LayerView& L, const float* wp, float* win, hipStream_t st);
// skip_near_uniform: return at once when every knot row is (nearly) equally spaced - the fp16 matrix kernel launched
// beside this one does the layer then
// amax_out: where the per-jet maxima of y are combined into (4 words, see kan_f16.hip), or nullptr
void launch_fwd_sparse(const LayerView& L, const float* a, const float* win, float* y, int64_t N, hipStream_t st,
                       bool skip_near_uniform = false, unsigned* amax_out = nullptr,
                       SeedSrc seed = SeedSrc());
// cot_out: where the per-jet maxima of ga are combined into (4 words), or nullptr
void launch_bwd_input_sparse(const LayerView& L, const float* a, const float* gy, const float* win, float* ga,
                             int64_t N, hipStream_t st, bool skip_near_uniform = false, unsigned* cot_out = nullptr);

// kan_f16.hip: hidden layers on the fp16 matrix cores with fp32 accuracy (two fp16 pieces per operand, three
// products), J = 4 jets; for layers whose knot rows are all (nearly) equally spaced - decided on the device
bool f16_eligible(int D, int D2, const LayerView& L);
// layers whose fp16 weight gradients take chunks of about 512 points (32 x 32 matrix tiles, no second register set)
bool f16_wide_weight_tiles(int D, int D2, const LayerView& L);
size_t f16_wplane_floats(const LayerView& L);
void launch_absmax(const float* w, int64_t n, unsigned* out, hipStream_t st);               // out[0] = bits of max |w|
void launch_jet_absmax(const float* t, int64_t elems, unsigned* out4, hipStream_t st);      // per-jet maxima of [elems][4]
void launch_pack_wplanes16(const LayerView& L, const float* wp, const unsigned* wmax, float* wpl, hipStream_t st, bool jets12 = false);
// amax_in: per-jet maxima of a (4 words); amax_out: where the per-jet maxima of y are combined into, or nullptr
void launch_fwd_f16(const LayerView& L, const float* a, const float* wpl, float* y, int64_t N, const unsigned* amax_in,
                    const unsigned* wmax, unsigned* amax_out, hipStream_t st);
// cot_in: per-jet maxima of gy (4 words); cot_out: where the per-jet maxima of ga are combined into, or nullptr
void launch_bwd_input_f16(const LayerView& L, const float* a, const float* gy, const float* wpl, float* ga, int64_t N,
                          const unsigned* cot_in, const unsigned* wmax, unsigned* cot_out, hipStream_t st);
// 12-jet layers (6 directions, 5 second derivatives; grid 10): same planes, per-slot maxima are 12 words;
// N must be a multiple of 8
bool f16_jets12_eligible(int D, int D2, const LayerView& L);
void launch_jet_absmax12(const float* t, int64_t elems, unsigned* out12, hipStream_t st);   // per-slot maxima of [elems][12]
void launch_fwd12_f16(const LayerView& L, const float* a, const float* wpl, float* y, int64_t N, const unsigned* amax_in,
                      const unsigned* wmax, hipStream_t st);
void launch_bwd_input12_f16(const LayerView& L, const float* a, const float* gy, const float* wpl, float* ga, int64_t N,
                            const unsigned* cot_in, const unsigned* wmax, hipStream_t st);
void launch_bwd_weight12_f16(const LayerView& L, const float* a, const float* gy, float* part, int64_t N16, int64_t chunk,
                             int nx, int ny, int nmain, const unsigned* act_in, const unsigned* cot_in, hipStream_t st);

// kan_mfma.hip: dense f32-MFMA path for wide hidden layers (J = 4 jets)
bool mfma_eligible(int D, int D2, const LayerView& L);
int mfma_weight_chunks(int64_t N, const LayerView& L);
bool mfma_skinny_weight_eligible(int D, int D2, const LayerView& L);   // 2-input layers: weight gradients only
int mfma_skinny_weight_chunks(int64_t N, const LayerView& L);
void launch_bwd_weight_mfma_skinny(const LayerView& L, const float* a, const float* gy, float* part, int nchunks,
                                   int64_t N, hipStream_t st);
void launch_fwd_mfma(const LayerView& L, const float* a, const float* wt, float* y, int64_t N, hipStream_t st);
void launch_bwd_input_mfma(const LayerView& L, const float* a, const float* gy, const float* wp, float* ga,
                           int64_t N, hipStream_t st);
// with_bf16x3: feature blocks with equally spaced knots go to the split-bf16 kernel, the rest stay fp32
bool bf16x3_weight_eligible(int D, int D2, const LayerView& L, int64_t N);
// act_in / cot_in (per-jet maxima of a and gy, 4 words each) given: the fp16 two-piece kernel instead of the split-bf16 one
// skip_layer (fp32 kernel only): write nothing when every knot row of the layer is (nearly) equally spaced
void launch_bwd_weight_mfma(const LayerView& L, const float* a, const float* gy, float* part, int nchunks,
                            int64_t N, hipStream_t st, bool with_bf16x3 = false, const unsigned* act_in = nullptr,
                            const unsigned* cot_in = nullptr, bool skip_layer = false);
void launch_bwd_weight_f16_blocks(const LayerView& L, const float* a, const float* gy, float* part, int64_t N16,
                                  int64_t chunk, int nx, int ny, int nmain, const unsigned* act_in,
                                  const unsigned* cot_in, hipStream_t st);

}  // namespace pdekan
