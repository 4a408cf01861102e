// C ABI of libpdekan (see include/pdekan.h): argument checks, workspace
// carving and the sequence of kernel launches of a layer / model / train step.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/pdekan.h"
#include "kan_launch.h"

// Built-in scratch of pdekan_layer_forward_noalloc (see scratch_of_current_device below).
static __device__ __align__(256) unsigned char scratch[PDEKAN_NOALLOC_SCRATCH_BYTES];

namespace pdekan {
// a refused raise of a kernel's dynamic-LDS limit, kept until the entry point's launch check reports it (the launch
// that follows would only say "invalid argument")
thread_local char g_lds_refused[200] = "";

void ensure_dynamic_lds(const void* kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return;
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t& g = granted[{dev, kernel}];
  if (g >= bytes) return;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) g = bytes;
  else
    snprintf(g_lds_refused, sizeof(g_lds_refused),
             "raising a kernel's dynamic LDS limit to %zu bytes on device %d failed: %s (first launch of a kernel shape "
             "inside stream capture?)", bytes, dev, hipGetErrorString(e));
}
}  // namespace pdekan

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (pdekan::g_lds_refused[0]) {
    const int rc = fail(PDEKAN_ERR_HIP, "%s: %s%s%s", what, pdekan::g_lds_refused, e != hipSuccess ? "; then " : "",
                        e != hipSuccess ? hipGetErrorString(e) : "");
    pdekan::g_lds_refused[0] = 0;
    return rc;
  }
  if (e != hipSuccess) return fail(PDEKAN_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return PDEKAN_OK;
}

// ---- optional per-kernel timing (HIP events on the launch stream) ----------
struct ProfRec { char name[48]; hipEvent_t a, b; };
struct ProfSum { char name[48]; double ms; long long launches; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::vector<ProfSum> g_prof_sums;

struct ProfScope {
  hipStream_t st; int idx = -1;
  ProfScope(hipStream_t s, const char* fmt, int layer) : st(s) {
    if (!g_prof_on) return;
    ProfRec r;
    snprintf(r.name, sizeof(r.name), fmt, layer);
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    g_prof_recs.push_back(r);
    idx = (int)g_prof_recs.size() - 1;
  }
  ~ProfScope() { if (idx >= 0) (void)hipEventRecord(g_prof_recs[idx].b, st); }
};

bool getenv_flag(const char* name) {
  const char* v = getenv(name);
  return v && v[0] == '1';
}

bool jet_supported(int d, int d2) { return (d == 0 && d2 == 0) || (d == 2 && d2 == 1) || (d == 6 && d2 == 5); }

int to_view(const PdekanLayer* l, pdekan::LayerView& v) {
  if (!l) return fail(PDEKAN_ERR_ARG, "layer is NULL");
  if (l->spline_order != 3) return fail(PDEKAN_ERR_ARG, "spline_order %d unsupported (cubic only)", l->spline_order);
  if (l->in_features < 1 || l->out_features < 1 || l->grid_size < 1 || l->grid_size > 64)
    return fail(PDEKAN_ERR_ARG, "bad layer shape in=%d out=%d grid=%d", l->in_features, l->out_features, l->grid_size);
  if (l->act_kind != PDEKAN_ACT_IDENTITY && l->act_kind != PDEKAN_ACT_SILU)
    return fail(PDEKAN_ERR_ARG, "act_kind %d unsupported", l->act_kind);
  if (!l->grid || !l->base_weight || !l->spline_weight) return fail(PDEKAN_ERR_ARG, "layer has NULL parameter pointers");
  v.I = l->in_features; v.O = l->out_features; v.G = l->grid_size; v.act = l->act_kind;
  v.grid = l->grid; v.base_w = l->base_weight; v.spline_w = l->spline_weight; v.scaler = l->spline_scaler;
  v.ginfo = nullptr;
  return PDEKAN_OK;
}

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// the four networks of pdekan_deeponet_step as the kernels' launcher takes them
struct DeepONetArgs {
  pdekan::LayerView layers[4][pdekan::kDonLayersMax];
  pdekan::SmallGrads grads[4][pdekan::kDonLayersMax];
  pdekan::DonNetView views[4];
};
int deeponet_args(const PdekanNet* nets, DeepONetArgs& A) {
  if (!nets) return fail(PDEKAN_ERR_ARG, "deeponet_step: nets is NULL");
  for (int k = 0; k < 4; ++k) {
    const PdekanNet& N = nets[k];
    if (!N.layers || N.n_layers < 1 || N.n_layers > pdekan::kDonLayersMax)
      return fail(PDEKAN_ERR_ARG, "deeponet_step: network %d has %d layers (1 .. %d)", k, N.n_layers, pdekan::kDonLayersMax);
    for (int l = 0; l < N.n_layers; ++l) {
      const int rc = to_view(&N.layers[l], A.layers[k][l]);
      if (rc) return rc;
      if (N.grads) {
        if (!N.grads[l].base_weight || !N.grads[l].spline_weight || (A.layers[k][l].scaler && !N.grads[l].spline_scaler))
          return fail(PDEKAN_ERR_ARG, "deeponet_step: gradient pointers missing");
        A.grads[k][l] = {N.grads[l].base_weight, N.grads[l].spline_weight, N.grads[l].spline_scaler};
      }
    }
    A.views[k] = {A.layers[k], N.grads ? A.grads[k] : nullptr, N.n_layers};
  }
  return PDEKAN_OK;
}

// bump allocator over the caller's workspace
struct Carver {
  char* base; size_t size, used = 0; bool ok = true;
  Carver(void* p, size_t n) : base(static_cast<char*>(p)), size(n) {}
  float* take(size_t floats) {
    const size_t bytes = align_up(floats * sizeof(float));
    if (used + bytes > size) { ok = false; return nullptr; }
    float* r = reinterpret_cast<float*>(base + used);
    used += bytes;
    return r;
  }
};

int weight_chunks(int D, int D2, const pdekan::LayerView& L, int64_t n) {
  if (pdekan::mfma_eligible(D, D2, L)) return pdekan::mfma_weight_chunks(n, L);
  if (pdekan::mfma_skinny_weight_eligible(D, D2, L)) return pdekan::mfma_skinny_weight_chunks(n, L);
  if (pdekan::seeded_first_layer(D, D2, L)) {               // four resident waves per SIMD's worth of chunks
    const int64_t want = 4096 / (L.O / 64), cap = (n + 127) / 128;
    return (int)(want < cap ? want : (cap < 1 ? 1 : cap));
  }
  return pdekan::generic_weight_chunks(n, L.I, L.O);
}

// 12-jet layers (6 directions, 5 second derivatives) whose gradients are taken in 5 passes of the 4-jet kernels
bool split_jets(int D, int D2) { return D == 6 && D2 == 5 && !getenv_flag("PDEKAN_DISABLE_JET_SPLIT"); }

size_t win_floats(int D, int D2, const pdekan::LayerView& L) {
  if (split_jets(D, D2) && pdekan::sparse_input_grad_eligible(2, 1, L)) return pdekan::sparse_table_floats(L);
  return pdekan::sparse_eligible(D, D2, L) ? pdekan::sparse_table_floats(L) : 0;
}

// layers of a layer-level call that the fp16 matrix kernels take (they need the sparse tables as their fallback)
bool layer_f16(int D, int D2, const pdekan::LayerView& L) {
  return (pdekan::f16_eligible(D, D2, L) && win_floats(D, D2, L) > 0) || pdekan::f16_jets12_eligible(D, D2, L);
}
// act / cot / w maxima of a layer-level call (256 bytes): a word per jet (4 or 12 used) at kStatAct and kStatCot, one at kStatW
constexpr size_t kLayerStatWords = 64;
constexpr int kStatAct = 0, kStatCot = 12, kStatW = 24;

// tables of a layer forward: packed weights (two orders), sparse tables, knot-row classes, fp16 planes and maxima
size_t layer_fwd_floats(int D, int D2, const pdekan::LayerView& L) {
  const size_t packed = align_up(L.packed_floats() * 4) / 4;
  size_t f = 2 * packed + align_up(win_floats(D, D2, L) * 4) / 4 + align_up((size_t)L.I * 4) / 4;
  if (layer_f16(D, D2, L)) f += align_up(pdekan::f16_wplane_floats(L) * 4) / 4 + kLayerStatWords;
  return f;
}

size_t jets12_part_chunks(const pdekan::LayerView& L, int64_t n, bool views);

size_t layer_part_floats(int D, int D2, const pdekan::LayerView& L, int64_t n) {
  size_t chunks = (size_t)weight_chunks(D, D2, L, n) + 1;
  if (pdekan::f16_jets12_eligible(D, D2, L)) {
    const size_t c12 = jets12_part_chunks(L, n, false);
    if (c12 > chunks) chunks = c12;
  }
  return chunks * L.packed_floats();
}

size_t layer_ws_floats(int D, int D2, const pdekan::LayerView& L, int64_t n) {
  return layer_fwd_floats(D, D2, L) + align_up(layer_part_floats(D, D2, L, n) * 4) / 4;
}

// What the fp16 matrix kernels of a layer need beside the layer itself: its W planes, the per-jet maxima of its
// input jets / of the cotangents of its outputs (4 words each, float bit patterns), the maximum of its packed
// weights, and where to combine the maxima of what it produces (nullptr = nobody needs them).
struct F16Io {
  const float* wpl = nullptr;
  const unsigned* act_in = nullptr;
  const unsigned* cot_in = nullptr;
  const unsigned* wmax = nullptr;
  unsigned* act_out = nullptr;
  unsigned* cot_out = nullptr;
};

// act_out (with a layer the sparse kernel takes): where that kernel leaves the per-jet maxima of y for an fp16
// matrix layer above
// seed (model-level calls, layer 0): the layer's input jets a are still to be built from the points; the kernels that
// can do that themselves (and leave the jets at a for the backward pass) save the seeding launch
int run_forward(int D, int D2, const pdekan::LayerView& L, const float* a, const float* wt, const float* win,
                float* y, int64_t n, hipStream_t st, int li = 0, const F16Io* h = nullptr, unsigned* act_out = nullptr,
                const pdekan::SeedSrc* seed = nullptr) {
  pdekan::SeedSrc fused;                      // what the forward kernel is given (empty: it reads a)
  if (seed) {
    const bool f16_route = h && h->wpl;
    const bool sparse_route = !f16_route && win && pdekan::sparse_eligible(D, D2, L);
    const bool generic_route = !f16_route && !sparse_route && !pdekan::mfma_eligible(D, D2, L);
    if ((sparse_route && pdekan::fwd_sparse_takes_seed(L)) || (generic_route && pdekan::fwd_generic_takes_seed(D, D2, L))) {
      fused = *seed;
    } else {
      ProfScope ps(st, "seed_jets", 0);
      pdekan::launch_seed_jets(D, D2, seed->x, seed->sn, seed->sd, seed->a_out, n, L.I, st);
    }
  }
  if (h && h->wpl && D == 6) {
    // 12 jets: the fp16 matrix kernel on whole groups of 8 points when every knot row is (nearly) equally spaced, else
    // the streaming kernel (decided on the device); the last n % 8 points always stream
    const int64_t n8 = n / 8 * 8;
    const size_t J = 12;
    if (n8) {
      {
        ProfScope ps(st, "L%d.fwd.mfma_f16x2", li);
        pdekan::launch_fwd12_f16(L, a, h->wpl, y, n8, h->act_in, h->wmax, st);
      }
      ProfScope ps(st, "L%d.fwd.standby", li);        // (returns at once unless a knot row is far from equal spacing)
      pdekan::launch_fwd_generic(D, D2, L, a, wt, y, n8, st, true);
    }
    if (n > n8) {
      ProfScope ps(st, "L%d.fwd.stream", li);
      pdekan::launch_fwd_generic(D, D2, L, a + (size_t)n8 * L.I * J, wt, y + (size_t)n8 * L.O * J, n - n8, st);
    }
  } else if (h && h->wpl && win) {
    // both on the same grid: the fp16 matrix kernel runs when every knot row is (nearly) equally spaced, else the
    // sparse kernel does (decided on the device)
    ProfScope ps(st, "L%d.fwd.mfma_f16x2", li);
    pdekan::launch_fwd_f16(L, a, h->wpl, y, n, h->act_in, h->wmax, h->act_out, st);
    pdekan::launch_fwd_sparse(L, a, win, y, n, st, true, h->act_out);
  } else if (win && pdekan::sparse_eligible(D, D2, L)) {
    ProfScope ps(st, "L%d.fwd.sparse", li);
    pdekan::launch_fwd_sparse(L, a, win, y, n, st, false, act_out, fused);
  } else if (pdekan::mfma_eligible(D, D2, L)) {
    ProfScope ps(st, "L%d.fwd.mfma", li);
    pdekan::launch_fwd_mfma(L, a, wt, y, n, st);
  } else {
    ProfScope ps(st, "L%d.fwd.stream", li);
    if (!pdekan::launch_fwd_generic(D, D2, L, a, wt, y, n, st, false, act_out, fused))
      return fail(PDEKAN_ERR_ARG, "jet shape (%d,%d) unsupported", D, D2);
  }
  return check_launch("layer forward");
}

int run_backward(int D, int D2, const pdekan::LayerView& L, const float* a, const float* gy, const float* wp,
                 const float* win, float* ga, float* part, const PdekanLayerGrad* g, int accumulate, int64_t n,
                 hipStream_t st, int li = 0, const F16Io* h = nullptr, bool seeded_input = false) {
  const bool mf = pdekan::mfma_eligible(D, D2, L);
  if (g) {
    if (!g->base_weight || !g->spline_weight || (L.scaler && !g->spline_scaler))
      return fail(PDEKAN_ERR_ARG, "gradient pointers missing");
    const int nch = weight_chunks(D, D2, L, n);
    if (mf && pdekan::bf16x3_weight_eligible(D, D2, L, n) && h && h->wpl && h->act_in && h->cot_in) {
      ProfScope ps(st, "L%d.bwd_weight.mfma_f16x2", li);
      pdekan::launch_bwd_weight_mfma(L, a, gy, part, nch, n, st, true, h->act_in, h->cot_in);
    } else if (mf && pdekan::bf16x3_weight_eligible(D, D2, L, n)) {
      ProfScope ps(st, "L%d.bwd_weight.mfma_bf16x3", li);
      pdekan::launch_bwd_weight_mfma(L, a, gy, part, nch, n, st, true);
    } else if (mf) {
      ProfScope ps(st, "L%d.bwd_weight.mfma", li);
      pdekan::launch_bwd_weight_mfma(L, a, gy, part, nch, n, st);
    } else if (pdekan::mfma_skinny_weight_eligible(D, D2, L)) {
      ProfScope ps(st, "L%d.bwd_weight.mfma_skinny", li);
      pdekan::launch_bwd_weight_mfma_skinny(L, a, gy, part, nch, n, st);
    } else {
      ProfScope ps(st, "L%d.bwd_weight.stream", li);
      if (!pdekan::launch_bwd_weight_generic(D, D2, L, a, gy, part, nch, n, st, false, seeded_input))
        return fail(PDEKAN_ERR_ARG, "jet shape (%d,%d) unsupported", D, D2);
    }
    ProfScope ps(st, "L%d.reduce_unpack", li);
    pdekan::launch_reduce_unpack(L, part, nch, part + (size_t)nch * L.packed_floats(), g->base_weight, g->spline_weight,
                                 L.scaler ? g->spline_scaler : nullptr, accumulate, st);
  }
  if (ga) {
    if (h && h->wpl && win) {
      // both on the same grid: the fp16 matrix kernel runs when every knot row is (nearly) equally spaced, else the
      // sparse kernel does (decided on the device)
      ProfScope ps(st, "L%d.bwd_input.mfma_f16x2", li);
      // (the matrix kernel walks whole groups of 16 points; the last n % 16 points always go to the sparse kernel)
      const int64_t n16 = n / 16 * 16;
      if (n16) {
        pdekan::launch_bwd_input_f16(L, a, gy, h->wpl, ga, n16, h->cot_in, h->wmax, h->cot_out, st);
        pdekan::launch_bwd_input_sparse(L, a, gy, win, ga, n16, st, true, h->cot_out);
      }
      if (n > n16) {
        const size_t J = 4;
        pdekan::launch_bwd_input_sparse(L, a + (size_t)n16 * L.I * J, gy + (size_t)n16 * L.O * J, win,
                                        ga + (size_t)n16 * L.I * J, n - n16, st, false, h->cot_out);
      }
    } else if (win && pdekan::sparse_input_grad_eligible(D, D2, L)) {
      ProfScope ps(st, "L%d.bwd_input.sparse", li);
      pdekan::launch_bwd_input_sparse(L, a, gy, win, ga, n, st);
    } else if (mf) {
      ProfScope ps(st, "L%d.bwd_input.mfma", li);
      pdekan::launch_bwd_input_mfma(L, a, gy, wp, ga, n, st);
    } else {
      ProfScope ps(st, "L%d.bwd_input.stream", li);
      if (!pdekan::launch_bwd_input_generic(D, D2, L, a, gy, wp, ga, n, st))
        return fail(PDEKAN_ERR_ARG, "jet shape (%d,%d) unsupported", D, D2);
    }
  }
  return check_launch("layer backward");
}

// Weight gradients of a 12-jet layer (6 directions, 5 second derivatives) as 5 passes of the 4-jet matrix kernels
// over gathered views (see gather_jets4_kernel): for layers those kernels take, in the fused step.
bool split_weight_eligible(int D, int D2, const pdekan::LayerView& L, int64_t n) {
  return split_jets(D, D2) && n >= 16 && (L.G == 10 || L.G == 20) && pdekan::mfma_eligible(2, 1, L);
}
constexpr int kSplitPasses = 5;

// The split passes of one layer: per pass one gather of the input jets and of the cotangents, the 4-jet
// weight-gradient kernels into consecutive groups of chunks (one reduction at the end) and, where the sparse
// input-gradient kernel takes the layer, its 4-jet input gradients scattered back into the 12-jet tensor.
// Returns through *ga_done whether the input gradients were produced here.
int run_backward_split(const pdekan::LayerView& L, const float* a, const float* gy, const float* win, float* ga,
                       float* const* view, float* part, const PdekanLayerGrad* g, int accumulate, int64_t n,
                       hipStream_t st, int li, bool* ga_done) {
  if (!g->base_weight || !g->spline_weight || (L.scaler && !g->spline_scaler))
    return fail(PDEKAN_ERR_ARG, "gradient pointers missing");
  const int nch = pdekan::mfma_weight_chunks(n, L);
  const bool bf = pdekan::bf16x3_weight_eligible(2, 1, L, n);
  const bool with_ga = ga && win && pdekan::sparse_input_grad_eligible(2, 1, L);
  for (int k = 0; k < kSplitPasses; ++k) {
    {
      ProfScope ps(st, "L%d.gather_views", li);
      pdekan::launch_gather_jets4(a, view[0], n * L.I, k, true, st);
      pdekan::launch_gather_jets4(gy, view[1], n * L.O, k, k == 0, st);
    }
    {
      ProfScope ps(st, bf ? "L%d.bwd_weight.mfma_bf16x3" : "L%d.bwd_weight.mfma", li);
      pdekan::launch_bwd_weight_mfma(L, view[0], view[1], part + (size_t)k * nch * L.packed_floats(), nch, n, st, bf);
    }
    if (with_ga) {
      {
        ProfScope ps(st, "L%d.bwd_input.sparse", li);
        pdekan::launch_bwd_input_sparse(L, view[0], view[1], win, view[2], n, st);
      }
      ProfScope ps(st, "L%d.scatter_views", li);
      pdekan::launch_scatter_jets4(view[2], ga, n * L.I, k, st);
    }
  }
  *ga_done = with_ga;
  ProfScope ps(st, "L%d.reduce_unpack", li);
  const int total = kSplitPasses * nch;
  pdekan::launch_reduce_unpack(L, part, total, part + (size_t)total * L.packed_floats(), g->base_weight,
                               g->spline_weight, L.scaler ? g->spline_scaler : nullptr, accumulate, st);
  return check_launch("layer backward (split passes)");
}

// 12-jet layers the fp16 matrix kernels take in the fused step: with the split passes (and the sparse input-gradient
// kernel under them) as what runs instead when some knot row is not (nearly) equally spaced
bool jets12_f16(int D, int D2, const pdekan::LayerView& L) {
  return pdekan::f16_jets12_eligible(D, D2, L) && split_jets(D, D2) && pdekan::mfma_eligible(2, 1, L) &&
         pdekan::sparse_input_grad_eligible(2, 1, L);
}

// chunks of partial weight gradients of such a layer on n points: [matrix kernel: nch][last n % 16 points: 1]
// [what runs instead of the matrix kernel: 5 * nch (split passes, with views) or the streaming kernel's chunks] and
// one chunk of scratch for the sum
int jets12_other_chunks(const pdekan::LayerView& L, int64_t n16, int nch, bool views) {
  if (!n16) return 0;
  return views ? kSplitPasses * nch : pdekan::generic_weight_chunks(n16, L.I, L.O);
}
size_t jets12_part_chunks(const pdekan::LayerView& L, int64_t n, bool views) {
  const int64_t n16 = n / 16 * 16;
  const int nch = n16 ? pdekan::mfma_weight_chunks(n16, L) : 0;
  return (size_t)nch + jets12_other_chunks(L, n16, nch, views) + 2;
}

// Backward of a 12-jet layer on the fp16 matrix kernels.  Points [0, n16): the matrix kernels when every knot row is
// (nearly) equally spaced, else the split passes (view given: the fused step) or the streaming kernels (layer-level
// calls) - both are launched, the device settles which one works, and the reduction sums the chunks of the one that
// did.  The last n % 16 points take the streaming kernels.
// h: W planes and the per-slot maxima of a and gy (12 words each).
int run_backward_jets12(const pdekan::LayerView& L, const float* a, const float* gy, const float* wp, const float* win,
                        float* ga, float* const* view, float* part, const PdekanLayerGrad* g, int accumulate, int64_t n,
                        hipStream_t st, int li, const F16Io* h) {
  if (g && (!g->base_weight || !g->spline_weight || (L.scaler && !g->spline_scaler)))
    return fail(PDEKAN_ERR_ARG, "gradient pointers missing");
  const size_t J = 12, pf = L.packed_floats();
  const int64_t n16 = n / 16 * 16, tail = n - n16;
  const int nch = n16 ? pdekan::mfma_weight_chunks(n16, L) : 0;
  const pdekan::JetSkip skip = {L.ginfo, L.I};
  float* part_tail = part + (size_t)nch * pf;
  float* part_split = part_tail + pf;
  if (n16) {
    if (g) {
      ProfScope ps(st, "L%d.bwd_weight.mfma_f16x2", li);
      int64_t chunk = (n16 + nch - 1) / nch;
      chunk = (chunk + 15) / 16 * 16;
      pdekan::launch_bwd_weight12_f16(L, a, gy, part, n16, chunk, L.I / 32, L.O / 64, nch, h->act_in, h->cot_in, st);
    }
    if (ga) {
      ProfScope ps(st, "L%d.bwd_input.mfma_f16x2", li);
      pdekan::launch_bwd_input12_f16(L, a, gy, h->wpl, ga, n16, h->cot_in, h->wmax, st);
    }
    ProfScope ps(st, "L%d.bwd.standby", li);           // (split passes / streaming kernels: return at once as above)
    if (!view) {
      if (g) pdekan::launch_bwd_weight_generic(6, 5, L, a, gy, part_split, jets12_other_chunks(L, n16, nch, false), n16, st, true);
      if (ga) pdekan::launch_bwd_input_generic(6, 5, L, a, gy, wp, ga, n16, st, true);
    }
    for (int k = 0; view && k < kSplitPasses; ++k) {
      pdekan::launch_gather_jets4(a, view[0], n16 * L.I, k, true, st, skip);
      pdekan::launch_gather_jets4(gy, view[1], n16 * L.O, k, k == 0, st, skip);
      if (g)
        pdekan::launch_bwd_weight_mfma(L, view[0], view[1], part_split + (size_t)k * nch * pf, nch, n16, st, false, nullptr,
                                       nullptr, true);
      if (ga) {
        pdekan::launch_bwd_input_sparse(L, view[0], view[1], win, view[2], n16, st, true);
        pdekan::launch_scatter_jets4(view[2], ga, n16 * L.I, k, st, skip);
      }
    }
  }
  if (tail) {
    ProfScope ps(st, "L%d.bwd.stream", li);
    const float* at = a + (size_t)n16 * L.I * J;
    const float* gt = gy + (size_t)n16 * L.O * J;
    if (g) pdekan::launch_bwd_weight_generic(6, 5, L, at, gt, part_tail, 1, tail, st);
    if (ga) pdekan::launch_bwd_input_generic(6, 5, L, at, gt, wp, ga + (size_t)n16 * L.I * J, tail, st);
  }
  if (g) {
    ProfScope ps(st, "L%d.reduce_unpack", li);
    const int t = tail ? 1 : 0;
    pdekan::ChunkSelect sel = {L.ginfo, L.I, 0, nch + t, tail ? nch : nch + 1, t + jets12_other_chunks(L, n16, nch, view != nullptr)};
    if (!n16) sel = {L.ginfo, L.I, nch, 1, nch, 1};
    float* sum_buf = part + (jets12_part_chunks(L, n, view != nullptr) - 1) * pf;
    pdekan::launch_reduce_unpack(L, part, 0, sum_buf, g->base_weight, g->spline_weight, L.scaler ? g->spline_scaler : nullptr,
                                 accumulate, st, sel);
  }
  return check_launch("layer backward (12 jets)");
}

struct ModelPlan {
  std::vector<pdekan::LayerView> L;
  int D, D2, J;
  int64_t mb;           // points per micro-batch
  int maxw;             // widest activation
  size_t part_floats;   // largest partial buffer
  size_t view_floats;   // one gathered 4-jet view of the widest activation, when some layer takes the split passes
};

int make_plan(const PdekanLayer* layers, int n_layers, int64_t n, int D, int D2, int64_t micro_batch, ModelPlan& P) {
  if (!layers || n_layers < 1) return fail(PDEKAN_ERR_ARG, "no layers");
  if (n < 1) return fail(PDEKAN_ERR_ARG, "n must be positive");
  if (!jet_supported(D, D2)) return fail(PDEKAN_ERR_ARG, "jet shape (%d,%d) unsupported", D, D2);
  P.L.resize(n_layers);
  P.D = D; P.D2 = D2; P.J = 1 + D + D2;
  P.mb = (micro_batch <= 0 || micro_batch > n) ? n : micro_batch;
  P.maxw = 0; P.part_floats = 0; P.view_floats = 0;
  for (int l = 0; l < n_layers; ++l) {
    int rc = to_view(&layers[l], P.L[l]);
    if (rc) return rc;
    if (l > 0 && P.L[l].I != P.L[l - 1].O) return fail(PDEKAN_ERR_ARG, "layer %d input width mismatch", l);
    if (P.L[l].I > P.maxw) P.maxw = P.L[l].I;
    if (P.L[l].O > P.maxw) P.maxw = P.L[l].O;
    size_t pf = (size_t)(weight_chunks(D, D2, P.L[l], P.mb) + 1) * P.L[l].packed_floats();
    if (split_weight_eligible(D, D2, P.L[l], P.mb)) {
      const size_t ps = (size_t)(kSplitPasses * pdekan::mfma_weight_chunks(P.mb, P.L[l]) + 1) * P.L[l].packed_floats();
      if (ps > pf) pf = ps;                  // a last micro-batch under 16 points takes the streaming kernel
      P.view_floats = 1;
    }
    if (jets12_f16(D, D2, P.L[l])) {
      const size_t ps = jets12_part_chunks(P.L[l], P.mb, true) * P.L[l].packed_floats();
      if (ps > pf) pf = ps;
      P.view_floats = 1;
    }
    if (pf > P.part_floats) P.part_floats = pf;
  }
  if (P.view_floats) P.view_floats = (size_t)P.mb * P.maxw * 4;
  {
    const pdekan::LayerView& H = P.L.back();
    if (pdekan::head_eligible(D, H)) {
      const size_t pf = (size_t)(pdekan::head_blocks(P.mb, H) + 1) * H.packed_floats();
      if (pf > P.part_floats) P.part_floats = pf;
    }
  }
  return PDEKAN_OK;
}

// Maxima the fp16 matrix kernels scale their operands by, one slot of a word per jet (4 or 12 used) each: per layer
// boundary l = 0 .. nl the per-jet maxima of the jets entering layer l (act) and of the cotangents of those jets (cot),
// per layer the maximum of its packed weights (w).  act and cot are cleared per micro-batch, w once per call.
struct ModelStats {
  unsigned* base; int nl;
  static constexpr int kSlot = 12;
  static size_t words(int nl) { return (size_t)(2 * kSlot + 4) * (nl + 1); }
  unsigned* act(int l) const { return base + kSlot * l; }
  unsigned* cot(int l) const { return base + kSlot * (nl + 1) + kSlot * l; }
  unsigned* w(int l) const { return base + 2 * kSlot * (nl + 1) + 4 * l; }
  size_t per_batch_bytes() const { return (size_t)2 * kSlot * (nl + 1) * 4; }
  size_t bytes() const { return words(nl) * 4; }
};

struct ModelWs {
  std::vector<float*> wp, wt, win, ginfo, act;   // act[l] = input jets of layer l; act[n_layers] = output jets
  std::vector<float*> wpl;                       // W planes of the layers the fp16 matrix kernels take, else nullptr
  unsigned* stats;                               // maxima (float bit patterns), see ModelStats
  float* g[2];
  float* view[3];       // 4-jet views of a layer's input jets, cotangents and input gradients (split passes), else nullptr
  float* part;
  float* loss_part;
};

constexpr int kLossBlocks = 1024;        // blocks of bs_residual_kernel (unfused residual + loss)
// one float per block of whichever kernel writes the loss partials
constexpr int kLossPartials = kLossBlocks > pdekan::kHeadBlocksMax ? kLossBlocks : pdekan::kHeadBlocksMax;
static_assert(kLossPartials >= pdekan::kHeadBlocksMax && kLossPartials >= kLossBlocks, "loss partials cover every block");

bool carve(const ModelPlan& P, void* ws, size_t ws_bytes, ModelWs& W, size_t* need) {
  Carver c(ws, ws ? ws_bytes : 0);
  const size_t nl = P.L.size();
  size_t total = 0;
  auto take = [&](size_t floats) -> float* {
    total += align_up(floats * 4);
    return ws ? c.take(floats) : nullptr;
  };
  W.wp.resize(nl); W.wt.resize(nl); W.win.resize(nl); W.ginfo.resize(nl); W.act.resize(nl + 1); W.wpl.resize(nl);
  W.stats = reinterpret_cast<unsigned*>(take(ModelStats::words((int)nl)));
  for (size_t l = 0; l < nl; ++l) {
    W.wp[l] = take(P.L[l].packed_floats());
    W.wt[l] = take(P.L[l].packed_floats());
    W.ginfo[l] = take(P.L[l].I);
    const size_t wf = win_floats(P.D, P.D2, P.L[l]);
    W.win[l] = wf ? take(wf) : nullptr;
    if (!ws && wf) W.win[l] = reinterpret_cast<float*>(1);   // sizing pass: only the total matters
    const bool f16 = wf && (pdekan::f16_eligible(P.D, P.D2, P.L[l]) || jets12_f16(P.D, P.D2, P.L[l]));
    W.wpl[l] = f16 ? take(pdekan::f16_wplane_floats(P.L[l])) : nullptr;
  }
  for (size_t l = 0; l <= nl; ++l) {
    const int w = (l < nl) ? P.L[l].I : P.L[nl - 1].O;
    W.act[l] = take((size_t)P.mb * w * P.J);
  }
  W.g[0] = take((size_t)P.mb * P.maxw * P.J);
  W.g[1] = take((size_t)P.mb * P.maxw * P.J);
  W.view[0] = P.view_floats ? take(P.view_floats) : nullptr;
  W.view[1] = P.view_floats ? take(P.view_floats) : nullptr;
  W.view[2] = P.view_floats ? take(P.view_floats) : nullptr;
  W.part = take(P.part_floats);
  W.loss_part = take(kLossPartials);
  if (need) *need = total;
  return ws ? c.ok : true;
}

// Per call: knot-row classes, packed weights and the tables / planes derived from them, for every layer.
int pack_model(ModelPlan& P, const ModelWs& W, const ModelStats& S, hipStream_t st) {
  const int nl = (int)P.L.size();
  // one launch: every layer's packed weights and knot-row classes, and the scale maxima cleared
  const bool packed = pdekan::launch_pack_model(P.L.data(), nl, W.wp.data(), W.wt.data(), W.ginfo.data(), S.base,
                                                (int)(S.bytes() / 4), st);
  if (!packed) {
    hipError_t e = hipMemsetAsync(S.base, 0, S.bytes(), st);
    if (e != hipSuccess) return fail(PDEKAN_ERR_HIP, "clearing the scale maxima: %s", hipGetErrorString(e));
  }
  for (int l = 0; l < nl; ++l) {
    if (!packed) pdekan::launch_pack_weights(P.L[l], W.wp[l], W.wt[l], st, W.ginfo[l]);   // (knot-row classes too)
    P.L[l].ginfo = W.ginfo[l];
    if (W.win[l]) pdekan::launch_pack_window(P.L[l], W.wp[l], W.win[l], st);
    if (W.wpl[l]) {
      pdekan::launch_absmax(W.wp[l], (int64_t)P.L[l].packed_floats(), S.w(l), st);
      pdekan::launch_pack_wplanes16(P.L[l], W.wp[l], S.w(l), W.wpl[l], st, P.D == 6);
    }
  }
  return check_launch("pack");
}

// Forward of layer l of a model on m points.  have_act: the per-jet maxima of the layer's input are already in
// S.act(l) (the layer before was an fp16 matrix kernel too and left them there).
int model_layer_forward(const ModelPlan& P, const ModelWs& W, const ModelStats& S, int l, float* dst, int64_t m,
                        hipStream_t st, bool* act_ready, const pdekan::SeedSrc* seed = nullptr) {
  const int nl = (int)P.L.size();
  F16Io io;
  const bool f16 = W.wpl[l] != nullptr;
  if (seed && f16) {                          // the maxima of the input jets are measured below: they have to exist
    ProfScope ps(st, "seed_jets", 0);
    pdekan::launch_seed_jets(P.D, P.D2, seed->x, seed->sn, seed->sd, seed->a_out, m, P.L[l].I, st);
    seed = nullptr;
  }
  if (f16) {
    if (!*act_ready) {
      ProfScope ps(st, "L%d.jet_absmax", l);
      if (P.D == 6) pdekan::launch_jet_absmax12(W.act[l], m * P.L[l].I, S.act(l), st);
      else pdekan::launch_jet_absmax(W.act[l], m * P.L[l].I, S.act(l), st);
    }
    io.wpl = W.wpl[l]; io.act_in = S.act(l); io.wmax = S.w(l);
    io.act_out = (P.D == 2 && l + 1 < nl && W.wpl[l + 1]) ? S.act(l + 1) : nullptr;   // (the 12-jet kernel leaves none)
  }
  // a layer the sparse kernel takes can leave the maxima of its outputs for an fp16 matrix layer above as well
  unsigned* sparse_out = nullptr;
  if (!f16 && l + 1 < nl && W.wpl[l + 1] && W.win[l] && pdekan::sparse_eligible(P.D, P.D2, P.L[l])) sparse_out = S.act(l + 1);
  // and so can a layer on the streaming kernel (the first layer of a 6-input model, or of a grid-20 model)
  const bool streams = !f16 && !(W.win[l] && pdekan::sparse_eligible(P.D, P.D2, P.L[l])) &&
                       !pdekan::mfma_eligible(P.D, P.D2, P.L[l]);
  if (streams && (P.D == 6 || P.D == 2) && l + 1 < nl && W.wpl[l + 1]) sparse_out = S.act(l + 1);
  *act_ready = (f16 && io.act_out != nullptr) || sparse_out != nullptr;
  return run_forward(P.D, P.D2, P.L[l], W.act[l], W.wt[l], W.win[l], dst, m, st, l, f16 ? &io : nullptr, sparse_out, seed);
}

// Tables of one layer-level call, carved from the caller's workspace in this order.
struct LayerTables { float *wp, *wt, *win, *ginfo, *wpl; unsigned* stats; };   // stats: see kStatAct, kStatCot, kStatW

bool carve_layer(Carver& c, int D, int D2, const pdekan::LayerView& L, bool with_win, LayerTables& T) {
  T.wp = c.take(L.packed_floats());
  T.wt = c.take(L.packed_floats());
  const bool f16 = layer_f16(D, D2, L);
  const size_t wf = (with_win || f16) ? win_floats(D, D2, L) : 0;
  T.win = wf ? c.take(wf) : nullptr;
  T.ginfo = c.take(L.I);
  T.wpl = f16 ? c.take(pdekan::f16_wplane_floats(L)) : nullptr;
  T.stats = f16 ? reinterpret_cast<unsigned*>(c.take(kLayerStatWords)) : nullptr;
  return c.ok;
}

int pack_layer(pdekan::LayerView& L, const LayerTables& T, hipStream_t st, int n_dir) {
  pdekan::launch_grid_info(L, T.ginfo, st);
  L.ginfo = T.ginfo;
  pdekan::launch_pack_weights(L, T.wp, T.wt, st);
  if (T.win) pdekan::launch_pack_window(L, T.wp, T.win, st);
  if (T.wpl) {
    if (hipMemsetAsync(T.stats, 0, kLayerStatWords * 4, st) != hipSuccess)
      return fail(PDEKAN_ERR_HIP, "clearing the scale maxima failed");
    pdekan::launch_absmax(T.wp, (int64_t)L.packed_floats(), T.stats + kStatW, st);
    pdekan::launch_pack_wplanes16(L, T.wp, T.stats + kStatW, T.wpl, st, n_dir == 6);
  }
  return PDEKAN_OK;
}

int layer_forward_impl(const PdekanLayer* layer, const float* x_jet, float* y_jet, int64_t n, int32_t n_dir,
                       int32_t n_second, void* ws, size_t ws_bytes, void* stream) {
  pdekan::LayerView L;
  int rc = to_view(layer, L);
  if (rc) return rc;
  if (!x_jet || !y_jet || n < 1) return fail(PDEKAN_ERR_ARG, "layer_forward: bad arguments");
  if (!jet_supported(n_dir, n_second)) return fail(PDEKAN_ERR_ARG, "jet shape (%d,%d) unsupported", n_dir, n_second);
  if (!ws || ((uintptr_t)ws & 255)) return fail(PDEKAN_ERR_WORKSPACE, "workspace NULL or not 256-byte aligned");
  Carver c(ws, ws_bytes);
  LayerTables T;
  if (!carve_layer(c, n_dir, n_second, L, true, T))
    return fail(PDEKAN_ERR_WORKSPACE, "workspace too small: %zu bytes given", ws_bytes);
  hipStream_t st = (hipStream_t)stream;
  rc = pack_layer(L, T, st, n_dir);
  if (rc) return rc;
  F16Io io;
  if (T.wpl) {
    if (n_dir == 6) pdekan::launch_jet_absmax12(x_jet, n * L.I, T.stats + kStatAct, st);
    else pdekan::launch_jet_absmax(x_jet, n * L.I, T.stats + kStatAct, st);
    io.wpl = T.wpl; io.act_in = T.stats + kStatAct; io.wmax = T.stats + kStatW;
  }
  return run_forward(n_dir, n_second, L, x_jet, T.wt, T.win, y_jet, n, st, 0, T.wpl ? &io : nullptr);
}

// ---- built-in scratch of pdekan_layer_forward_noalloc ------------------------------------------------------
// Device memory the library owns from load time on (no hipMalloc, ever): the packed weight tables of ONE layer
// forward.  Every GPU the library is loaded on holds its own copy of the array at its own address; `scratch` is
// only the host-side handle of the symbol, so the address is asked for once per device (with that device current)
// and host code uses that pointer, never the array's name.  Every table in it is rewritten in full by the packing
// kernels of each call, so nothing has to be cleared between calls.
constexpr size_t kScratchBytes = PDEKAN_NOALLOC_SCRATCH_BYTES;
constexpr int kMaxDevices = 64;
std::atomic<void*> g_scratch_ptr[kMaxDevices];

int scratch_of_current_device(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(PDEKAN_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= kMaxDevices) return fail(PDEKAN_ERR_ARG, "device index %d out of range", dev);
  void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));       // the same address whichever thread asks first
    if (e != hipSuccess || !p) return fail(PDEKAN_ERR_HIP, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
    if ((uintptr_t)p & 255) return fail(PDEKAN_ERR_HIP, "built-in scratch is not 256-byte aligned");
    g_scratch_ptr[dev].store(p, std::memory_order_release);
  }
  *out = p;
  return PDEKAN_OK;
}

}  // namespace

extern "C" {

int pdekan_version(void) { return 100; }

const char* pdekan_last_error(void) { return g_err; }

int pdekan_seed_jets(const float* x, int64_t stride_n, int64_t stride_d, int64_t n, int32_t d, int32_t n_dir,
                     int32_t n_second, float* jet, void* stream) {
  if (!x || !jet || n < 1 || d < 1) return fail(PDEKAN_ERR_ARG, "seed_jets: bad arguments");
  if (n_dir != 0 && n_dir != d) return fail(PDEKAN_ERR_ARG, "seed_jets: n_dir must be 0 or d");
  if (!pdekan::launch_seed_jets(n_dir, n_second, x, stride_n, stride_d, jet, n, d, (hipStream_t)stream))
    return fail(PDEKAN_ERR_ARG, "jet shape (%d,%d) unsupported", n_dir, n_second);
  return check_launch("seed_jets");
}

int pdekan_basis_dense(const float* grid, const float* x, int64_t n, int32_t in_features, int32_t grid_size,
                       int32_t spline_order, int32_t order, float* out, void* stream) {
  if (!grid || !x || !out || n < 1 || in_features < 1) return fail(PDEKAN_ERR_ARG, "basis_dense: bad arguments");
  if (spline_order < 0 || spline_order > 5) return fail(PDEKAN_ERR_ARG, "basis_dense: spline_order %d unsupported (0..5)", spline_order);
  if (grid_size < 1 || grid_size > 64) return fail(PDEKAN_ERR_ARG, "bad grid_size %d", grid_size);
  if (order < 0 || order > 6) return fail(PDEKAN_ERR_ARG, "derivative order %d: 0..6 (orders above the spline order give zeros)", order);
  if (spline_order == 3 && order > 3) {
    if (hipMemsetAsync(out, 0, (size_t)n * in_features * (grid_size + 3) * sizeof(float), (hipStream_t)stream) != hipSuccess)
      return fail(PDEKAN_ERR_HIP, "basis_dense: clearing the output failed");
    return PDEKAN_OK;
  }
  pdekan::launch_basis_dense(x, grid, out, n, in_features, grid_size, order, (hipStream_t)stream, spline_order);
  return check_launch("basis_dense");
}

int pdekan_fit_coefficients(const float* grid, const float* x, int64_t x_stride_n, int64_t x_stride_i, int64_t n,
                            int32_t in_features, int32_t out_features, int32_t grid_size, int32_t spline_order,
                            const float* y, int64_t y_stride_n, int64_t y_stride_i, int64_t y_stride_o,
                            const float* old_grid, const float* old_spline_weight, const float* old_spline_scaler,
                            float* coeff, void* stream) {
  if (!grid || !x || !coeff || n < 1 || in_features < 1 || out_features < 1)
    return fail(PDEKAN_ERR_ARG, "fit_coefficients: bad arguments");
  if (spline_order != 3) return fail(PDEKAN_ERR_ARG, "fit_coefficients: spline_order %d unsupported (cubic only)", spline_order);
  if (grid_size < 1 || grid_size > 20) return fail(PDEKAN_ERR_ARG, "fit_coefficients: grid_size %d unsupported (1..20)", grid_size);
  if (!y && (!old_grid || !old_spline_weight))
    return fail(PDEKAN_ERR_ARG, "fit_coefficients: neither samples y nor the layer on its old grid given");
  if (!pdekan::launch_fit_coefficients(x, x_stride_n, x_stride_i, n, in_features, out_features, grid_size, grid, y,
                                       y_stride_n, y_stride_i, y_stride_o, old_grid, old_spline_weight,
                                       old_spline_scaler, coeff, (hipStream_t)stream))
    return fail(PDEKAN_ERR_ARG, "fit_coefficients: grid_size %d unsupported", grid_size);
  return check_launch("fit_coefficients");
}

size_t pdekan_layer_workspace_bytes(const PdekanLayer* layer, int64_t n, int32_t n_dir, int32_t n_second) {
  pdekan::LayerView L;
  if (to_view(layer, L) || n < 1) return 0;
  return layer_ws_floats(n_dir, n_second, L, n) * sizeof(float) + 256;
}

int pdekan_layer_forward(const PdekanLayer* layer, const float* x_jet, float* y_jet, int64_t n, int32_t n_dir,
                         int32_t n_second, void* ws, size_t ws_bytes, void* stream) {
  return layer_forward_impl(layer, x_jet, y_jet, n, n_dir, n_second, ws, ws_bytes, stream);
}

int pdekan_layer_forward_noalloc(const PdekanLayer* layer, const float* x_jet, float* y_jet, int64_t n, int32_t n_dir,
                                 int32_t n_second, void* stream) {
  pdekan::LayerView L;
  int rc = to_view(layer, L);
  if (rc) return rc;
  if (!jet_supported(n_dir, n_second)) return fail(PDEKAN_ERR_ARG, "jet shape (%d,%d) unsupported", n_dir, n_second);
  // the packed tables do not depend on n: size them for one point
  const size_t need = layer_fwd_floats(n_dir, n_second, L) * sizeof(float);
  if (need > kScratchBytes)
    return fail(PDEKAN_ERR_WORKSPACE,
                "layer_forward_noalloc: a %d -> %d layer at grid %d needs %zu bytes of packed tables, the built-in "
                "scratch holds %zu (PDEKAN_NOALLOC_SCRATCH_BYTES); use pdekan_layer_forward with a workspace",
                L.I, L.O, L.G, need, kScratchBytes);
  void* ptr = nullptr;
  rc = scratch_of_current_device(&ptr);
  if (rc) return rc;
  return layer_forward_impl(layer, x_jet, y_jet, n, n_dir, n_second, ptr, kScratchBytes, stream);
}

int pdekan_layer_backward(const PdekanLayer* layer, const float* x_jet, const float* gy_jet, int64_t n,
                          int32_t n_dir, int32_t n_second, float* gx_jet, const PdekanLayerGrad* grads,
                          int32_t accumulate, void* ws, size_t ws_bytes, void* stream) {
  pdekan::LayerView L;
  int rc = to_view(layer, L);
  if (rc) return rc;
  if (!x_jet || !gy_jet || n < 1) return fail(PDEKAN_ERR_ARG, "layer_backward: bad arguments");
  if (!jet_supported(n_dir, n_second)) return fail(PDEKAN_ERR_ARG, "jet shape (%d,%d) unsupported", n_dir, n_second);
  if (!ws || ((uintptr_t)ws & 255)) return fail(PDEKAN_ERR_WORKSPACE, "workspace NULL or not 256-byte aligned");
  Carver c(ws, ws_bytes);
  LayerTables T;
  const bool tables_ok = carve_layer(c, n_dir, n_second, L, gx_jet != nullptr, T);
  float* part = c.take(layer_part_floats(n_dir, n_second, L, n));
  if (!tables_ok || !c.ok) return fail(PDEKAN_ERR_WORKSPACE, "workspace too small: %zu bytes given", ws_bytes);
  hipStream_t st = (hipStream_t)stream;
  rc = pack_layer(L, T, st, n_dir);
  if (rc) return rc;
  F16Io io;
  const bool f16 = T.wpl && n_dir == 2;
  if (T.wpl && n_dir == 6) {
    pdekan::launch_jet_absmax12(x_jet, n * L.I, T.stats + kStatAct, st);
    pdekan::launch_jet_absmax12(gy_jet, n * L.O, T.stats + kStatCot, st);
    io.wpl = T.wpl; io.act_in = T.stats + kStatAct; io.cot_in = T.stats + kStatCot; io.wmax = T.stats + kStatW;
    return run_backward_jets12(L, x_jet, gy_jet, T.wp, T.win, gx_jet, nullptr, part, grads, accumulate, n, st, 0, &io);
  }
  if (f16) {
    pdekan::launch_jet_absmax(x_jet, n * L.I, T.stats + kStatAct, st);
    pdekan::launch_jet_absmax(gy_jet, n * L.O, T.stats + kStatCot, st);
    io.wpl = T.wpl; io.act_in = T.stats + kStatAct; io.cot_in = T.stats + kStatCot; io.wmax = T.stats + kStatW;
  }
  return run_backward(n_dir, n_second, L, x_jet, gy_jet, T.wp, T.win, gx_jet, part, grads, accumulate, n, st, 0,
                      f16 ? &io : nullptr);
}

size_t pdekan_model_workspace_bytes(const PdekanLayer* layers, int32_t n_layers, int64_t n, int32_t n_dir,
                                    int32_t n_second, int64_t micro_batch) {
  ModelPlan P;
  if (make_plan(layers, n_layers, n, n_dir, n_second, micro_batch, P)) return 0;
  ModelWs W;
  size_t need = 0;
  carve(P, nullptr, 0, W, &need);
  if (n_dir == 2 && pdekan::small_step_eligible(P.L.data(), (int)P.L.size())) {   // (the two-launch step's partial sums)
    const size_t small = pdekan::small_step_part_floats(P.L.data(), (int)P.L.size(), n) * sizeof(float);
    if (small > need) need = small;
  }
  return need + 256;
}

int pdekan_model_forward_jet(const PdekanLayer* layers, int32_t n_layers, const float* x, int64_t stride_n,
                             int64_t stride_d, int64_t n, int32_t n_dir, int32_t n_second, float* out_jet,
                             int64_t micro_batch, void* ws, size_t ws_bytes, void* stream) {
  ModelPlan P;
  int rc = make_plan(layers, n_layers, n, n_dir, n_second, micro_batch, P);
  if (rc) return rc;
  if (!x || !out_jet) return fail(PDEKAN_ERR_ARG, "model_forward_jet: NULL buffers");
  if (n_dir != 0 && n_dir != P.L[0].I) return fail(PDEKAN_ERR_ARG, "n_dir must be 0 or the input width");
  if (!ws || ((uintptr_t)ws & 255)) return fail(PDEKAN_ERR_WORKSPACE, "workspace NULL or not 256-byte aligned");
  ModelWs W;
  if (!carve(P, ws, ws_bytes, W, nullptr)) return fail(PDEKAN_ERR_WORKSPACE, "workspace too small: %zu bytes given", ws_bytes);
  hipStream_t st = (hipStream_t)stream;
  const int nl = (int)P.L.size();
  const ModelStats S = {W.stats, nl};
  rc = pack_model(P, W, S, st);
  if (rc) return rc;
  const int out_w = P.L[nl - 1].O;
  for (int64_t p0 = 0; p0 < n; p0 += P.mb) {
    const int64_t m = (n - p0 < P.mb) ? (n - p0) : P.mb;
    const pdekan::SeedSrc seed = {x + p0 * stride_n, stride_n, stride_d, W.act[0]};   // layer 0 builds its input jets
    if (p0 > 0 && hipMemsetAsync(S.base, 0, S.per_batch_bytes(), st) != hipSuccess)
      return fail(PDEKAN_ERR_HIP, "clearing the scale maxima failed");
    bool act_ready = false;
    for (int l = 0; l < nl; ++l) {
      float* dst = (l == nl - 1) ? out_jet + p0 * out_w * P.J : W.act[l + 1];
      rc = model_layer_forward(P, W, S, l, dst, m, st, &act_ready, l == 0 ? &seed : nullptr);
      if (rc) return rc;
    }
  }
  return check_launch("model_forward_jet");
}

int pdekan_bs_step(const PdekanLayer* layers, int32_t n_layers, const float* x, int64_t stride_n,
                   int64_t stride_d, const float* target, int64_t n, int64_t n_global, float sigma, float rate,
                   float* loss_out, float* res_out, const PdekanLayerGrad* grads, int32_t accumulate,
                   int64_t micro_batch, void* ws, size_t ws_bytes, void* stream) {
  if (!layers || n_layers < 1) return fail(PDEKAN_ERR_ARG, "no layers");
  const int d = layers[0].in_features;
  ModelPlan P;
  int rc = make_plan(layers, n_layers, n, d, d - 1, micro_batch, P);
  if (rc) return rc;
  if (!x) return fail(PDEKAN_ERR_ARG, "bs_step: x is NULL");
  if (P.L.back().O != 1) return fail(PDEKAN_ERR_ARG, "bs_step: the last layer must have one output");
  if (n_global < n) return fail(PDEKAN_ERR_ARG, "bs_step: n_global < n");
  if (!ws || ((uintptr_t)ws & 255)) return fail(PDEKAN_ERR_WORKSPACE, "workspace NULL or not 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nl = (int)P.L.size();
  const float inv_n = 1.0f / (float)n_global;
  if (P.D == 2 && pdekan::small_step_eligible(P.L.data(), nl)) {
    // a small model: every layer in one launch, thread = point (no per-point workspace, no micro-batches)
    // (the size the query function names: the same workspace serves pdekan_model_forward_jet)
    if (pdekan_model_workspace_bytes(layers, n_layers, n, d, d - 1, micro_batch) - 256 > ws_bytes)
      return fail(PDEKAN_ERR_WORKSPACE, "workspace too small: %zu bytes given", ws_bytes);
    pdekan::SmallGrads sg[pdekan::kSmallLayersMax];
    for (int l = 0; grads && l < nl; ++l) {
      if (!grads[l].base_weight || !grads[l].spline_weight || (P.L[l].scaler && !grads[l].spline_scaler))
        return fail(PDEKAN_ERR_ARG, "gradient pointers missing");
      sg[l] = {grads[l].base_weight, grads[l].spline_weight, grads[l].spline_scaler};
    }
    ProfScope ps(st, "small_step", 0);
    pdekan::launch_small_step(P.L.data(), grads ? sg : nullptr, nl, x, stride_n, stride_d, target, n, sigma, rate, inv_n,
                              res_out, loss_out, static_cast<float*>(ws), accumulate, st);
    return check_launch("bs_step (small model)");
  }
  ModelWs W;
  if (!carve(P, ws, ws_bytes, W, nullptr)) return fail(PDEKAN_ERR_WORKSPACE, "workspace too small: %zu bytes given", ws_bytes);
  const ModelStats S = {W.stats, nl};
  {
    ProfScope ps(st, "pack_weights", 0);
    rc = pack_model(P, W, S, st);
    if (rc) return rc;
  }
  bool first = true;
  for (int64_t p0 = 0; p0 < n; p0 += P.mb) {
    const int64_t m = (n - p0 < P.mb) ? (n - p0) : P.mb;
    const float* xs = x + p0 * stride_n;
    const pdekan::SeedSrc seed = {xs, stride_n, stride_d, W.act[0]};   // layer 0 builds its input jets (or is given them first)
    const pdekan::LayerView& H = P.L[nl - 1];
    const bool fused_head = pdekan::head_eligible(P.D, H) && !getenv_flag("PDEKAN_DISABLE_FUSED_HEAD");
    const int n_fwd = fused_head ? nl - 1 : nl;
    if (p0 > 0 && hipMemsetAsync(S.base, 0, S.per_batch_bytes(), st) != hipSuccess)
      return fail(PDEKAN_ERR_HIP, "clearing the scale maxima failed");
    bool act_ready = false;
    for (int l = 0; l < n_fwd; ++l) {
      rc = model_layer_forward(P, W, S, l, W.act[l + 1], m, st, &act_ready, l == 0 ? &seed : nullptr);
      if (rc) return rc;
    }
    if (n_fwd == 0) {                           // a one-layer model on the fused head: nobody else builds its input jets
      ProfScope ps(st, "seed_jets", 0);
      pdekan::launch_seed_jets(P.D, P.D2, xs, stride_n, stride_d, W.act[0], m, P.L[0].I, st);
    }
    const float* tgt = target ? target + p0 : nullptr;
    float* res = res_out ? res_out + p0 : nullptr;
    const int acc_flag = (accumulate || !first) ? 1 : 0;
    int top = nl - 1;            // first layer the generic backward loop still has to do
    int cur = 0;
    bool head_left_cot = false;
    if (fused_head) {
      const int nblocks = pdekan::head_blocks(m, H);
      // the head's input cotangents feed layer nl-2: leave their per-jet maxima if that layer runs on the fp16 kernels
      unsigned* head_cot = (grads && nl > 1 && W.wpl[nl - 2]) ? S.cot(nl - 1) : nullptr;
      head_left_cot = head_cot != nullptr;
      {
        ProfScope ps(st, "L%d.head_fused", nl - 1);
        pdekan::launch_head_bs(P.D, H, W.act[nl - 1], xs, stride_n, stride_d, tgt, W.wp[nl - 1], sigma, rate, inv_n,
                               res, (grads && nl > 1) ? W.g[0] : nullptr, grads ? W.part : nullptr, W.loss_part,
                               nblocks, m, st, head_cot);
        // (with gradients asked for, the loss reduction rides in the launch that reduces the head's gradients)
        if (loss_out && !grads) pdekan::launch_loss_finish(W.loss_part, nblocks, inv_n, loss_out, first ? 0 : 1, st);
      }
      if (grads) {
        const PdekanLayerGrad& g = grads[nl - 1];
        if (!g.base_weight || !g.spline_weight || (H.scaler && !g.spline_scaler))
          return fail(PDEKAN_ERR_ARG, "gradient pointers missing");
        ProfScope ps(st, "L%d.reduce_unpack", nl - 1);
        pdekan::LossJob lj;
        if (loss_out) { lj.part = W.loss_part; lj.nparts = nblocks; lj.inv_n = inv_n; lj.out = loss_out; lj.accumulate = first ? 0 : 1; }
        pdekan::launch_reduce_unpack(H, W.part, nblocks, W.part + (size_t)nblocks * H.packed_floats(), g.base_weight,
                                     g.spline_weight, H.scaler ? g.spline_scaler : nullptr, acc_flag, st, {}, lj);
      }
      top = nl - 2;
    } else {
      int nblocks = (int)((m + 255) / 256);
      if (nblocks > kLossBlocks) nblocks = kLossBlocks;
      ProfScope ps_res(st, "bs_residual_loss", 0);
      if (!pdekan::launch_bs_residual(P.D, xs, stride_n, stride_d, W.act[nl], tgt, sigma, rate, inv_n, res,
                                      grads ? W.g[0] : nullptr, W.loss_part, nblocks, m, st))
        return fail(PDEKAN_ERR_ARG, "bs_step: %d input columns unsupported (2 or 6)", d);
      if (loss_out) pdekan::launch_loss_finish(W.loss_part, nblocks, inv_n, loss_out, first ? 0 : 1, st);
    }
    if (grads) {
      bool cot_ready = head_left_cot;
      for (int l = top; l >= 0; --l) {
        float* gnext = (l > 0) ? W.g[cur ^ 1] : nullptr;
        if (P.D == 6 && W.wpl[l]) {
          if (!cot_ready) {
            ProfScope ps(st, "L%d.jet_absmax", l);
            pdekan::launch_jet_absmax12(W.g[cur], m * P.L[l].O, S.cot(l + 1), st);
          }
          cot_ready = false;                  // (the 12-jet kernels leave no maxima of what they write)
          F16Io io;
          io.wpl = W.wpl[l]; io.act_in = S.act(l); io.cot_in = S.cot(l + 1); io.wmax = S.w(l);
          rc = run_backward_jets12(P.L[l], W.act[l], W.g[cur], W.wp[l], W.win[l], gnext, W.view, W.part, &grads[l], acc_flag,
                                   m, st, l, &io);
        } else if (W.view[0] && split_weight_eligible(P.D, P.D2, P.L[l], m)) {
          bool ga_done = false;
          rc = run_backward_split(P.L[l], W.act[l], W.g[cur], W.win[l], gnext, W.view, W.part, &grads[l], acc_flag, m, st,
                                  l, &ga_done);
          if (!rc && gnext && !ga_done)
            rc = run_backward(P.D, P.D2, P.L[l], W.act[l], W.g[cur], W.wp[l], W.win[l], gnext, W.part, nullptr, acc_flag,
                              m, st, l);
        } else {
          F16Io io;
          const bool f16 = W.wpl[l] != nullptr && P.D == 2;
          if (f16) {
            // per-jet maxima of the cotangents of this layer's outputs: left by the layer above when it ran on the
            // fp16 matrix kernels too, else measured here
            if (!cot_ready) {
              ProfScope ps(st, "L%d.jet_absmax", l);
              pdekan::launch_jet_absmax(W.g[cur], m * P.L[l].O, S.cot(l + 1), st);
            }
            io.wpl = W.wpl[l]; io.act_in = S.act(l); io.cot_in = S.cot(l + 1); io.wmax = S.w(l);
            io.cot_out = (gnext && l > 0 && W.wpl[l - 1]) ? S.cot(l) : nullptr;
          }
          cot_ready = f16 && io.cot_out != nullptr;
          rc = run_backward(P.D, P.D2, P.L[l], W.act[l], W.g[cur], W.wp[l], W.win[l], gnext, W.part, &grads[l], acc_flag,
                            m, st, l, f16 ? &io : nullptr, l == 0);     // (W.act[0]: the seed jets)
        }
        if (rc) return rc;
        cur ^= 1;
      }
    }
    first = false;
  }
  return check_launch("bs_step");
}

size_t pdekan_deeponet_workspace_bytes(const PdekanNet* nets, int64_t sensors, int64_t n_bc, int64_t n_pde) {
  DeepONetArgs A;
  if (deeponet_args(nets, A)) return 0;
  if (const char* why = pdekan::deeponet_step_refusal(A.views, sensors, n_bc, n_pde)) {
    fail(PDEKAN_ERR_ARG, "deeponet_step: %s", why);
    return 0;
  }
  return pdekan::deeponet_step_ws_floats(A.views, sensors, n_bc, n_pde) * sizeof(float) + 256;
}

int pdekan_deeponet_step(const PdekanNet* nets, const float* u1, const float* u2, int64_t sensors, const float* x_bc,
                         const float* t_bc, const float* out_bc, int64_t n_bc, const float* x_pde, const float* t_pde,
                         const float* out_pde, int64_t n_pde, float sigma, float rate, float bc_weight, float* losses,
                         float* s_pred, float* res_out, void* ws, size_t ws_bytes, void* stream) {
  DeepONetArgs A;
  int rc = deeponet_args(nets, A);
  if (rc) return rc;
  if (!u1 || !u2 || !x_bc || !t_bc || !out_bc || !x_pde || !t_pde || !out_pde)
    return fail(PDEKAN_ERR_ARG, "deeponet_step: NULL buffers");
  if (const char* why = pdekan::deeponet_step_refusal(A.views, sensors, n_bc, n_pde))
    return fail(PDEKAN_ERR_ARG, "deeponet_step: %s", why);
  if (!ws || ((uintptr_t)ws & 255)) return fail(PDEKAN_ERR_WORKSPACE, "workspace NULL or not 256-byte aligned");
  if (pdekan::deeponet_step_ws_floats(A.views, sensors, n_bc, n_pde) * sizeof(float) > ws_bytes)
    return fail(PDEKAN_ERR_WORKSPACE, "workspace too small: %zu bytes given", ws_bytes);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, "deeponet_step", 0);
  if (!pdekan::launch_deeponet_step(A.views, u1, u2, sensors, x_bc, t_bc, out_bc, n_bc, x_pde, t_pde, out_pde, n_pde,
                                    sigma, rate, bc_weight, losses, s_pred, res_out, static_cast<float*>(ws), st))
    return fail(PDEKAN_ERR_ARG, "deeponet_step: refused");
  return check_launch("deeponet_step");
}

int pdekan_profile_begin(void) {
  for (auto& r : g_prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof_recs.clear();
  g_prof_sums.clear();
  g_prof_on = true;
  return PDEKAN_OK;
}

int pdekan_profile_end(void) {
  g_prof_on = false;
  int rc = PDEKAN_OK;
  for (auto& r : g_prof_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess)
      rc = fail(PDEKAN_ERR_HIP, "profile: event timing failed for %s", r.name);
    size_t k = 0;
    for (; k < g_prof_sums.size(); ++k)
      if (!strcmp(g_prof_sums[k].name, r.name)) break;
    if (k == g_prof_sums.size()) {
      ProfSum s;
      strncpy(s.name, r.name, sizeof(s.name));
      s.ms = 0.0; s.launches = 0;
      g_prof_sums.push_back(s);
    }
    g_prof_sums[k].ms += ms;
    g_prof_sums[k].launches += 1;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof_recs.clear();
  return rc;
}

int pdekan_profile_count(void) { return (int)g_prof_sums.size(); }

int pdekan_profile_entry(int idx, char* name, size_t name_cap, double* total_ms, int64_t* launches) {
  if (idx < 0 || idx >= (int)g_prof_sums.size() || !name || name_cap == 0)
    return fail(PDEKAN_ERR_ARG, "profile_entry: bad index");
  strncpy(name, g_prof_sums[idx].name, name_cap - 1);
  name[name_cap - 1] = 0;
  if (total_ms) *total_ms = g_prof_sums[idx].ms;
  if (launches) *launches = g_prof_sums[idx].launches;
  return PDEKAN_OK;
}

}  // extern "C"
