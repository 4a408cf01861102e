/* pdekan - C ABI of the MI355X (gfx950) KAN layer / Black-Scholes PINN step.
 *
 * The reference (iqq-0906/PDE) has no native interface for this path: its only
 * boundary is the Python module `efficient_kan` (efficient_kan/__init__.py:1-3)
 * whose `KANLinear.forward` / `KAN.forward` (efficient_kan/kan.py:153-166,
 * :274-279) are differentiated by PyTorch autograd inside the scripts'
 * `residual_net` / `loss_res` (run1.py:120-149).  The entry points below are
 * what a binding for that path has to call; each comment names the reference
 * code it stands in for.
 *
 * Conventions
 *   - All pointers are DEVICE pointers owned by the caller (fp32, contiguous
 *     unless strides are passed).  The library allocates nothing: scratch comes
 *     in through `ws` / `ws_bytes` (query the *_workspace_bytes function first;
 *     256-byte aligned).
 *   - `stream` is a hipStream_t (NULL = default stream).  Kernels are enqueued
 *     asynchronously; no call synchronises.
 *   - Return value 0 = ok, negative = error; text via pdekan_last_error()
 *     (thread-local).  Nothing throws across the ABI.
 *   - A "jet" of a scalar is J = 1 + n_dir + n_second floats:
 *       [0] value, [1..n_dir] d/dx_d, [1+n_dir ..] d2/dx_d^2 for d < n_second.
 *     Jet tensors are laid out [N][features][J].  Supported (n_dir, n_second):
 *     (0,0) plain values, (2,1) one asset + time, (6,5) five assets + time.
 *   - spline_order must be 3 (every caller in the reference uses the default).
 *   - act_kind: 0 = identity (run1.py:478-482 pass nn.Identity), 1 = SiLU
 *     (module default, kan.py:17).
 */
#ifndef PDEKAN_H
#define PDEKAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDEKAN_OK 0
#define PDEKAN_ERR_ARG (-1)        /* bad shape / unsupported combination */
#define PDEKAN_ERR_WORKSPACE (-2)  /* ws too small or misaligned */
#define PDEKAN_ERR_HIP (-3)        /* a HIP call failed */

#define PDEKAN_ACT_IDENTITY 0
#define PDEKAN_ACT_SILU 1

/* Parameters and buffers of one KANLinear (kan.py:27-45):
 *   grid [in][grid_size+2*spline_order+1], base_weight [out][in],
 *   spline_weight [out][in][grid_size+spline_order],
 *   spline_scaler [out][in] or NULL (enable_standalone_scale_spline=False). */
typedef struct PdekanLayer {
  int32_t in_features;
  int32_t out_features;
  int32_t grid_size;
  int32_t spline_order;
  int32_t act_kind;
  int32_t reserved;
  const float* grid;
  const float* base_weight;
  const float* spline_weight;
  const float* spline_scaler;
} PdekanLayer;

/* Where the gradients of one layer's parameters go (same shapes as the
 * parameters; spline_scaler may be NULL when the layer has none). */
typedef struct PdekanLayerGrad {
  float* base_weight;
  float* spline_weight;
  float* spline_scaler;
} PdekanLayerGrad;

int pdekan_version(void);
const char* pdekan_last_error(void);

/* Seed the input jets of the first layer from points x[N][d] (element strides
 * given, so the permuted view of run1.py:101-106 needs no copy):
 * value = x, d/dx_d = 1 for feature d, second derivatives = 0. */
int pdekan_seed_jets(const float* x, int64_t stride_n, int64_t stride_d, int64_t n, int32_t d,
                     int32_t n_dir, int32_t n_second, float* jet, void* stream);

/* KANLinear.b_splines (kan.py:78-111), dense, and its derivatives with respect to x as autograd would form them:
 * out[N][in][grid_size + spline_order] = d^order B_c(x[n][i]) / dx^order for x[N][in] (contiguous), on the layer's
 * grid[in][grid_size + 2 * spline_order + 1]; spline_order 0..5, order 0..6 (half-open knot intervals, zero outside
 * the extended grid; derivatives beyond the spline order are zero).  This is
 * the one piece of KANLinear.forward that torch.func transforms (jacrev / hessian, run1.py:122-124) cannot trace
 * through a library call: with it as an elementwise op that knows its own derivative, forward under those
 * transforms is two GEMMs around this kernel. */
int pdekan_basis_dense(const float* grid, const float* x, int64_t n, int32_t in_features, int32_t grid_size,
                       int32_t spline_order, int32_t order, float* out, void* stream);

/* KANLinear.curve2coeff (kan.py:113-143): coeff[out][in][grid_size + 3] = least-squares spline coefficients through
 * the samples (x[n][i], y[n][i][o]) on grid[in][grid_size + 7], per input feature, the minimal-norm solution where the
 * samples leave coefficients undetermined (torch.linalg.lstsq's, kan.py:131-133; rank cut at eps * max(n, coeffs)
 * like its default).  x and y take element strides (update_grid passes a permuted view, kan.py:178-180).
 * With y = NULL the samples are the layer's own edge functions on its OLD grid,
 *   y[n][i][o] = sum_c B_c(x[n][i]; old_grid) * old_spline_weight[o][i][c] * old_spline_scaler[o][i]  (scaler NULL: 1)
 * which is the refit update_grid does after moving the knots (kan.py:173-180, :215) without the [n][in][out] tensor.
 * Solved in fp64 (normal equations + Jacobi eigen-decomposition per feature); spline_order 3, grid_size <= 20.
 * grid and old_grid must not alias coeff; no workspace. */
int pdekan_fit_coefficients(const float* grid, const float* x, int64_t x_stride_n, int64_t x_stride_i, int64_t n,
                            int32_t in_features, int32_t out_features, int32_t grid_size, int32_t spline_order,
                            const float* y, int64_t y_stride_n, int64_t y_stride_i, int64_t y_stride_o,
                            const float* old_grid, const float* old_spline_weight, const float* old_spline_scaler,
                            float* coeff, void* stream);

/* KANLinear.forward (kan.py:153-166) on jets: y_jet[N][out][J] from
 * x_jet[N][in][J].  With (n_dir, n_second) = (0,0) this is exactly the
 * reference forward; otherwise the derivative slots carry what autograd /
 * jacrev / hessian would give for d y / d x_d and d2 y / d x_d^2. */
size_t pdekan_layer_workspace_bytes(const PdekanLayer* layer, int64_t n, int32_t n_dir, int32_t n_second);
int pdekan_layer_forward(const PdekanLayer* layer, const float* x_jet, float* y_jet, int64_t n,
                         int32_t n_dir, int32_t n_second, void* ws, size_t ws_bytes, void* stream);

/* pdekan_layer_forward without a workspace argument.  No counterpart in the reference: it is for callers that
 * can neither allocate device memory nor keep a workspace around - code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  The packed weight tables live in a device array built into the
 * library (PDEKAN_NOALLOC_SCRATCH_BYTES per GPU, held from library load to process end); the call allocates and
 * frees nothing.  A layer whose tables exceed that size is refused with PDEKAN_ERR_WORKSPACE (a 128 -> 128 layer
 * at grid 10 fits; 256 -> 256 at grid 20 does not).  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (different streams or threads) race; calls on one stream are ordered.  The first call on
 * a device looks the array's address up (hipGetSymbolAddress): make it outside graph capture.  The scratch used
 * is that of the CURRENT device (hipGetDevice): `stream` must belong to it, as must every pointer passed. */
#define PDEKAN_NOALLOC_SCRATCH_BYTES (8u << 20)
int pdekan_layer_forward_noalloc(const PdekanLayer* layer, const float* x_jet, float* y_jet, int64_t n,
                                 int32_t n_dir, int32_t n_second, void* stream);

/* What loss.backward() does through that forward (run1.py:225): given the
 * cotangent gy_jet[N][out][J], write gx_jet[N][in][J] (skipped if NULL) and
 * the parameter gradients (skipped if grads is NULL; added to the existing
 * contents if accumulate != 0). */
int pdekan_layer_backward(const PdekanLayer* layer, const float* x_jet, const float* gy_jet, int64_t n,
                          int32_t n_dir, int32_t n_second, float* gx_jet, const PdekanLayerGrad* grads,
                          int32_t accumulate, void* ws, size_t ws_bytes, void* stream);

/* KAN.forward (kan.py:274-279) on jets for a whole stack: points x[N][d] ->
 * out_jet[N][out_last][J] (operator_net + jacrev/hessian of run1.py:121-124 in
 * one pass).  micro_batch = points processed at a time (0 = all at once). */
size_t pdekan_model_workspace_bytes(const PdekanLayer* layers, int32_t n_layers, int64_t n,
                                    int32_t n_dir, int32_t n_second, int64_t micro_batch);
int pdekan_model_forward_jet(const PdekanLayer* layers, int32_t n_layers, const float* x, int64_t stride_n,
                             int64_t stride_d, int64_t n, int32_t n_dir, int32_t n_second, float* out_jet,
                             int64_t micro_batch, void* ws, size_t ws_bytes, void* stream);

/* One physics-informed training step without the optimiser update: forward,
 * Black-Scholes residual (run1.py:128; several assets: summed diagonal terms),
 * loss = sum((res - target)^2) / n_global (run1.py:148; n_global = n on one
 * GPU, the job-wide count on a shard) and every parameter gradient
 * (run1.py:225).  x[N][d] has the asset prices first and time last; target may
 * be NULL (zeros, run1.py:358); res_out (N floats) and loss_out (1 float) may
 * be NULL.  The last model layer must have one output. */
int pdekan_bs_step(const PdekanLayer* layers, int32_t n_layers, const float* x, int64_t stride_n,
                   int64_t stride_d, const float* target, int64_t n, int64_t n_global, float sigma,
                   float rate, float* loss_out, float* res_out, const PdekanLayerGrad* grads,
                   int32_t accumulate, int64_t micro_batch, void* ws, size_t ws_bytes, void* stream);

/* One training step of the physics-informed DeepONet the scripts build around four KANs, without the optimiser
 * update (class PI_DeepONet, run1.py:75-149; loop body run1.py:217-233; networks run1.py:478-482):
 *   v[r]      = model1(u1[r]) * model2(u2[r])            one value per sensor row, run1.py:92-98
 *   B         = model4(v as ONE row of `sensors` inputs)  run1.py:111-113 (the script repeats the row n times)
 *   s(x, t)   = sum_k B[k] * model5(x, t)[k]              run1.py:114-117
 *   bc loss   = mean((out_bc - s)^2)                      run1.py:134-141
 *   pde loss  = mean((out_pde - res)^2), res = s_t - sigma^2/2 x^2 s_xx - rate x s_x + rate s   run1.py:120-129, :145-149
 *   losses[3] = { pde loss + bc_weight * bc loss (run1.py:224), pde loss, bc loss }
 * and, if the networks carry gradient pointers, what loss.backward() (run1.py:225) leaves in every parameter's
 * gradient (overwritten, not accumulated).  nets[4] = model1, model2, model4, model5; u1 [sensors][in of model1] and
 * u2 [sensors][in of model2] contiguous; x, t, out of the boundary batch [n_bc] and of the collocation batch [n_pde]
 * contiguous.  s_pred [n_bc] and res_out [n_pde] may be NULL.  The step runs in four launches whatever the sizes;
 * it takes networks whose layers are no wider than 8 (any number of inputs to model4's first layer, one per sensor
 * row), at most 4 layers each, grid_size <= 20, at most 65536 points in all.  pdekan_deeponet_workspace_bytes
 * returns 0 for anything else (pdekan_last_error says what) and callers then run the networks layer by layer. */
typedef struct PdekanNet {
  const PdekanLayer* layers;
  const PdekanLayerGrad* grads;    /* one per layer, or NULL: losses only (all four networks alike) */
  int32_t n_layers;
  int32_t reserved;
} PdekanNet;
size_t pdekan_deeponet_workspace_bytes(const PdekanNet* nets, int64_t sensors, int64_t n_bc, int64_t n_pde);
int pdekan_deeponet_step(const PdekanNet* nets, const float* u1, const float* u2, int64_t sensors,
                         const float* x_bc, const float* t_bc, const float* out_bc, int64_t n_bc,
                         const float* x_pde, const float* t_pde, const float* out_pde, int64_t n_pde,
                         float sigma, float rate, float bc_weight, float* losses, float* s_pred, float* res_out,
                         void* ws, size_t ws_bytes, void* stream);

/* Optional per-kernel timing for benchmarks (no counterpart in the reference,
 * which only prints a whole-script wall time, run1.py:524-526).  Between
 * begin and end every kernel launched by the calls above is bracketed by HIP
 * events on the launch stream; end waits for them and sums the durations per
 * kernel name ("L1.fwd.mfma", "L0.bwd_weight.stream", "bs_residual_loss" ...).
 * The bracketing events serialise nothing but cost a little: do not time a
 * step with profiling on.  Not thread-safe; one profiling session at a time. */
int pdekan_profile_begin(void);
int pdekan_profile_end(void);
int pdekan_profile_count(void);
int pdekan_profile_entry(int idx, char* name, size_t name_cap, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* PDEKAN_H */
