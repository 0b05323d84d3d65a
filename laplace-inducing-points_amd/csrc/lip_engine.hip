// Op-tape interpreter + C ABI of the linearised-network engine (see include/lip.h).
//
// The Python host compiles a NetSpec into three tapes of lip_op_t (primal, tangent-forward,
// backward).  The engine resolves operand references against the bound device buffers and
// launches the HIP kernels on the caller's stream, chunking the probe dimension so that the
// tangent workspace fits the budget the caller allocated.  No allocation, no synchronisation
// and no host<->device copy happens inside a run (hipGraph-capturable).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "lip_internal.h"

namespace lip {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace lip

using namespace lip;

struct lip_engine {
  int64_t D = 0;
  int32_t n_img = 0, K = 0;
  std::vector<lip_op_t> tape[3];
  const float* theta = nullptr;
  const float* consts = nullptr;
  float* prim = nullptr;
  float* work = nullptr;
  int64_t work_pp = 0;      // workspace floats per probe
  int32_t max_chunk = 0;    // probes per chunk
  bool primal_done = false;
  // optional per-op timing (HIP events recorded on the launch stream)
  bool prof = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<int> ev_kind;   // kind of the op bracketed by events (2i, 2i+1)
};

namespace {

struct RunCtx {
  const lip_engine* e;
  const float* V;   // chunk-shifted
  float* Y;
  float* H;         // HEAD space, chunk-shifted
  int P;            // probes in this chunk
  int head_mode;
  float head_c;
  hipStream_t st;
  bool rows = false;   // lip_vjp_rows: Y holds one row per (probe, example); no reduction crosses examples
  // summed products (lip_ggn_vp / lip_vjp): a weight gradient that reduces all rows in one block may WRITE
  // y = s acc + alpha v instead of adding to an initialised block (the initialisation then skips its parameters)
  bool fuse = false;
  float alpha = 0.f;
};

inline float* resolve(const RunCtx& c, const lip_ref_t& r) {
  switch (r.space) {
    case LIP_SP_THETA: return const_cast<float*>(c.e->theta) + r.off;
    case LIP_SP_CONST: return const_cast<float*>(c.e->consts) + r.off;
    case LIP_SP_PRIM: return c.e->prim + r.off;
    case LIP_SP_WORK: return c.e->work + r.off * (int64_t)c.e->max_chunk;
    case LIP_SP_VIN: return c.V ? const_cast<float*>(c.V) + r.off : nullptr;
    case LIP_SP_YOUT: return c.Y ? c.Y + r.off : nullptr;
    case LIP_SP_HEAD: return c.H ? c.H + r.off : nullptr;
    default: return nullptr;
  }
}

// lip_vjp_rows: the parameter reductions an op would fuse (bias / BN cotangents: column sums over ALL rows of a
// probe) are taken per example instead, by a segmented reduce over the op's freshly written output.
struct RowReds { float* red0; float* red1; const float* xhat; };

#define RUN_CHECK(expr, what)                                                                  \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(_e)); return LIP_ERR_HIP; } \
  } while (0)

// Segmented (per-example) parameter reductions of lip_vjp_rows, taken over an op's output tensor [P][n][rows][N].
int rows_reduce(const RunCtx& c, const float* out, long long out_ps, int n_img, int rows, int N, const float* xhat,
                float* red0, long long red0_ps, float* red1, long long red1_ps) {
  if (!red0 && !red1) return LIP_OK;
  ReduceP r;
  memset(&r, 0, sizeof(r));
  r.g = out; r.g_ps = out_ps; r.R = rows; r.N = N; r.xhat = xhat;
  r.red0 = red0; r.red1 = red1;
  r.nseg = n_img; r.red_seg = red0 ? red0_ps : red1_ps;
  r.red0_ps = red0_ps * n_img; r.red1_ps = red1_ps * n_img;
  if (N <= 0 || N > 8192 || (red1 && !xhat)) { set_error("per-example reduce: bad operands"); return LIP_ERR_ARG; }
  RUN_CHECK(launch_reduce(r, c.P, c.st), "per-example reduce launch");
  return LIP_OK;
}

int check_space(const RunCtx& c, const lip_ref_t& r, const char* what) {
  if (r.space == LIP_SP_NONE) return LIP_OK;
  if (resolve(c, r) == nullptr) { set_error("op operand '%s' refers to an unbound space %d", what, r.space); return LIP_ERR_STATE; }
  return LIP_OK;
}

// resolved parameter block of a WGRAD op in this context (also queried by the initialisation plan of lip_ggn_vp / lip_vjp)
int make_wgrad(const RunCtx& c, const lip_op_t& op, WgradP& p) {
  const lip_seg_t& g = op.seg[0];
  p = WgradP();
  p.a = resolve(c, g.a);
  p.IH = g.IH; p.IW = g.IW; p.C = g.C; p.KH = g.KH; p.KW = g.KW;
  p.stride = g.stride; p.pad_h = g.pad_h; p.pad_w = g.pad_w;
  p.g = resolve(c, g.b); p.g_ps = g.b.pstride;
  p.R = op.n_img * op.OH * op.OW; p.OHW = op.OH * op.OW; p.OW = op.OW; p.N = op.N;
  p.M = g.KH * g.KW * g.C;
  if (op.OH <= 0 || op.OW <= 0) { set_error("WGRAD: bad geometry"); return LIP_ERR_ARG; }
  p.dOHW = FastDiv((unsigned)p.OHW); p.dOW = FastDiv((unsigned)p.OW);
  p.y = resolve(c, op.out); p.y_ps = op.out.pstride;
  p.scale = resolve(c, op.scale);
  p.ksplit = op.ksplit > 0 ? op.ksplit : 0;           // 0: the launcher picks the row split for the probe count
  if (c.rows) {
    p.ksplit = op.n_img; p.seg_rows = p.OHW; p.seg_ys = p.y_ps; p.y_ps *= op.n_img;
  }
  if (!p.a || !p.g || !p.y || p.R <= 0 || p.N <= 0 || p.M <= 0) { set_error("WGRAD: bad operands"); return LIP_ERR_ARG; }
  if ((long long)p.R * p.N >= (1ll << 31) || (long long)op.n_img * p.IH * p.IW * p.C >= (1ll << 31) || (long long)p.M * p.N >= (1ll << 31)) {
    set_error("WGRAD: a tensor of this binding has 2^31 or more elements; bind fewer examples per engine (ExampleChunkedGGN)");
    return LIP_ERR_ARG;
  }
  if ((p.C & 3) == 0 && (((uintptr_t)p.a) & 15)) { set_error("WGRAD: activations not 16-byte aligned"); return LIP_ERR_ARG; }
  if (c.fuse && !c.rows && op.out.space == LIP_SP_YOUT && wgrad_will_overwrite(p, c.P)) {
    p.overwrite = 1;
    p.v = c.V ? c.V + op.out.off : nullptr; p.v_ps = op.out.pstride; p.alpha = c.alpha;
  }
  return LIP_OK;
}

int run_op(const RunCtx& c, const lip_op_t& op) {
  switch (op.kind) {
    case LIP_OP_IGEMM: {
      if (op.nseg < 1 || op.nseg > 3) { set_error("IGEMM: nseg=%d", op.nseg); return LIP_ERR_ARG; }
      IgemmP p = IgemmP();
      p.nseg = op.nseg;
      for (int s = 0; s < op.nseg; ++s) {
        const lip_seg_t& g = op.seg[s];
        int rc;
        if ((rc = check_space(c, g.a, "seg.a")) || (rc = check_space(c, g.b, "seg.b"))) return rc;
        SegP& q = p.seg[s];
        q.a = resolve(c, g.a); q.a_ps = g.a.pstride;
        q.b = resolve(c, g.b); q.b_ps = g.b.pstride;
        q.IH = g.IH; q.IW = g.IW; q.C = g.C; q.KH = g.KH; q.KW = g.KW;
        q.stride = g.stride; q.pad_h = g.pad_h; q.pad_w = g.pad_w; q.mode = g.mode;
        q.Ktot = g.KH * g.KW * g.C;
        q.b_trans = (g.flags & LIP_SEG_B_TRANS) ? 1 : 0;
        q.dC = FastDiv((unsigned)(g.C > 0 ? g.C : 1)); q.dKW = FastDiv((unsigned)(g.KW > 0 ? g.KW : 1));
        if (g.mode == 0) { q.mul = g.stride; q.sgn = 1; q.off_h = -g.pad_h; q.off_w = -g.pad_w; q.mask = 0; q.sh = 0; }
        else { q.mul = 1; q.sgn = -1; q.off_h = g.pad_h; q.off_w = g.pad_w; q.mask = g.stride - 1; q.sh = (g.stride == 2) ? 1 : 0; }
        if (!q.a || !q.b || q.Ktot <= 0 || q.stride <= 0) { set_error("IGEMM: bad segment %d", s); return LIP_ERR_ARG; }
        if ((q.C & 3) == 0 && ((((uintptr_t)q.a) & 15) || (q.a_ps & 3))) { set_error("IGEMM: segment %d activations not 16-byte aligned", s); return LIP_ERR_ARG; }
      }
      p.R = op.n_img * op.OH * op.OW; p.OHW = op.OH * op.OW; p.OW = op.OW; p.N = op.N;
      if (op.OH <= 0 || op.OW <= 0) { set_error("IGEMM: bad geometry"); return LIP_ERR_ARG; }
      p.dOHW = FastDiv((unsigned)p.OHW); p.dOW = FastDiv((unsigned)p.OW);
      p.out = resolve(c, op.out); p.out_ps = op.out.pstride;
      p.no_ksplit = op.out.space == LIP_SP_PRIM;
      p.scale = resolve(c, op.scale);
      p.e0 = resolve(c, op.e0); p.e0_ps = op.e0.pstride;
      p.e1 = resolve(c, op.e1); p.e1_ps = op.e1.pstride;
      p.xhat = resolve(c, op.xhat);
      p.res = resolve(c, op.res); p.res_ps = op.res.pstride;
      p.dphi = resolve(c, op.dphi);
      p.red0 = resolve(c, op.red0); p.red0_ps = op.red0.pstride;
      p.red1 = resolve(c, op.red1); p.red1_ps = op.red1.pstride;
      p.xhat2 = resolve(c, op.xhat2);
      if (!p.out || p.R <= 0 || p.N <= 0) { set_error("IGEMM: bad output"); return LIP_ERR_ARG; }
      {   // the kernels index tensors with 32-bit arithmetic: refuse bindings that do not fit instead of wrapping
        const long long lim = 1ll << 31;
        bool fits = (long long)p.R * p.N < lim;
        for (int s = 0; s < op.nseg; ++s)
          fits = fits && (long long)op.n_img * p.seg[s].IH * p.seg[s].IW * p.seg[s].C < lim && (long long)p.seg[s].Ktot * p.N < lim;
        if (!fits) { set_error("IGEMM: a tensor of this binding has 2^31 or more elements; bind fewer examples per engine (ExampleChunkedGGN)"); return LIP_ERR_ARG; }
      }
      if (p.e1 && !p.xhat) { set_error("IGEMM: e1 without xhat"); return LIP_ERR_ARG; }
      if (p.red1 && !p.xhat2) { set_error("IGEMM: red1 without xhat2"); return LIP_ERR_ARG; }
      if (c.rows && (p.red0 || p.red1)) {
        float* r0 = p.red0; float* r1 = p.red1;
        p.red0 = nullptr; p.red1 = nullptr;
        RUN_CHECK(launch_igemm(p, c.P, c.st), "igemm launch");
        return rows_reduce(c, p.out, p.out_ps, op.n_img, p.OHW, p.N, p.xhat2, r0, p.red0_ps, r1, p.red1_ps);
      }
      RUN_CHECK(launch_igemm(p, c.P, c.st), "igemm launch");
      return LIP_OK;
    }
    case LIP_OP_WGRAD: {
      WgradP p;
      const int rc = make_wgrad(c, op, p);
      if (rc) return rc;
      RUN_CHECK(launch_wgrad(p, c.P, c.st), "wgrad launch");
      return LIP_OK;
    }
    case LIP_OP_REDUCE: {
      ReduceP p;
      memset(&p, 0, sizeof(p));
      p.g = resolve(c, op.seg[0].a); p.g_ps = op.seg[0].a.pstride;
      p.R = op.n_img * op.OH * op.OW; p.N = op.N;
      p.xhat = resolve(c, op.xhat2);
      p.red0 = resolve(c, op.red0); p.red0_ps = op.red0.pstride;
      p.red1 = resolve(c, op.red1); p.red1_ps = op.red1.pstride;
      if (!p.g || p.N <= 0 || p.N > 8192 || (p.red1 && !p.xhat)) { set_error("REDUCE: bad operands"); return LIP_ERR_ARG; }
      if (c.rows) {
        p.R = op.OH * op.OW; p.nseg = op.n_img; p.red_seg = p.red0 ? p.red0_ps : p.red1_ps;
        p.red0_ps *= op.n_img; p.red1_ps *= op.n_img;
      }
      RUN_CHECK(launch_reduce(p, c.P, c.st), "reduce launch");
      return LIP_OK;
    }
    case LIP_OP_POOL_FWD:
    case LIP_OP_POOL_BWD: {
      PoolP p;
      memset(&p, 0, sizeof(p));
      p.in = resolve(c, op.seg[0].a); p.in_ps = op.seg[0].a.pstride;
      p.out = resolve(c, op.out); p.out_ps = op.out.pstride;
      p.n = op.n_img; p.HW = op.OH * op.OW; p.C = op.N; p.inv = op.fscale;
      p.dphi = resolve(c, op.dphi);
      p.xhat = resolve(c, op.xhat2);
      p.red0 = resolve(c, op.red0); p.red0_ps = op.red0.pstride;
      p.red1 = resolve(c, op.red1); p.red1_ps = op.red1.pstride;
      if (!p.in || !p.out || p.C <= 0 || p.C > 8192 || (p.red1 && !p.xhat)) { set_error("POOL: bad operands"); return LIP_ERR_ARG; }
      if (op.kind == LIP_OP_POOL_FWD) RUN_CHECK(launch_pool_fwd(p, c.P, c.st), "pool_fwd launch");
      else if (c.rows && (p.red0 || p.red1)) {
        float* r0 = p.red0; float* r1 = p.red1;
        p.red0 = nullptr; p.red1 = nullptr;
        RUN_CHECK(launch_pool_bwd(p, c.P, c.st), "pool_bwd launch");
        return rows_reduce(c, p.out, p.out_ps, p.n, p.HW, p.C, p.xhat, r0, p.red0_ps, r1, p.red1_ps);
      }
      else RUN_CHECK(launch_pool_bwd(p, c.P, c.st), "pool_bwd launch");
      return LIP_OK;
    }
    case LIP_OP_MAXPOOL_PRIMAL:
    case LIP_OP_MAXPOOL_FWD:
    case LIP_OP_MAXPOOL_BWD: {
      const lip_seg_t& g = op.seg[0];
      MaxPoolP p;
      memset(&p, 0, sizeof(p));
      p.in = resolve(c, g.a); p.in_ps = g.a.pstride;
      p.out = resolve(c, op.out); p.out_ps = op.out.pstride;
      p.amax_w = resolve(c, op.aux0); p.amax = p.amax_w;
      p.n = op.n_img; p.IH = g.IH; p.IW = g.IW; p.OH = op.OH; p.OW = op.OW; p.C = op.N;
      p.KH = g.KH; p.KW = g.KW; p.stride = g.stride; p.pad_h = g.pad_h; p.pad_w = g.pad_w;
      p.dphi = resolve(c, op.dphi);
      p.xhat = resolve(c, op.xhat2);
      p.red0 = resolve(c, op.red0); p.red0_ps = op.red0.pstride;
      p.red1 = resolve(c, op.red1); p.red1_ps = op.red1.pstride;
      if (!p.in || !p.out || p.C <= 0 || p.C > 8192 || p.stride <= 0 || (p.red1 && !p.xhat)) { set_error("MAXPOOL: bad operands"); return LIP_ERR_ARG; }
      if (op.kind == LIP_OP_MAXPOOL_PRIMAL) RUN_CHECK(launch_maxpool_primal(p, c.st), "maxpool_primal launch");
      else if (op.kind == LIP_OP_MAXPOOL_FWD) RUN_CHECK(launch_maxpool_fwd(p, c.P, c.st), "maxpool_fwd launch");
      else if (c.rows && (p.red0 || p.red1)) {
        float* r0 = p.red0; float* r1 = p.red1;
        p.red0 = nullptr; p.red1 = nullptr;
        RUN_CHECK(launch_maxpool_bwd(p, c.P, c.st), "maxpool_bwd launch");
        return rows_reduce(c, p.out, p.out_ps, p.n, p.IH * p.IW, p.C, p.xhat, r0, p.red0_ps, r1, p.red1_ps);
      }
      else RUN_CHECK(launch_maxpool_bwd(p, c.P, c.st), "maxpool_bwd launch");
      return LIP_OK;
    }
    case LIP_OP_PRIMAL_POST: {
      PrimalPostP p;
      memset(&p, 0, sizeof(p));
      p.z = resolve(c, op.seg[0].a);
      p.a = resolve(c, op.out); p.dphi = resolve(c, op.out2); p.xhat = resolve(c, op.out3);
      p.bias = resolve(c, op.e0);
      p.gamma = resolve(c, op.e1); p.beta = resolve(c, op.scale);
      p.mean = resolve(c, op.aux0); p.rstd = resolve(c, op.aux1);
      p.res = resolve(c, op.res);
      p.count = (long long)op.n_img * op.OH * op.OW * op.N; p.N = op.N; p.act = op.act;
      if (!p.z || !p.a || (p.gamma && (!p.beta || !p.mean || !p.rstd))) { set_error("PRIMAL_POST: bad operands"); return LIP_ERR_ARG; }
      RUN_CHECK(launch_primal_post(p, c.st), "primal_post launch");
      return LIP_OK;
    }
    case LIP_OP_SOFTMAX: {
      const float* in = resolve(c, op.seg[0].a);
      float* pr = resolve(c, op.out); float* sq = resolve(c, op.out2);
      if (!in || !pr || !sq) { set_error("SOFTMAX: bad operands"); return LIP_ERR_ARG; }
      RUN_CHECK(launch_softmax(in, pr, sq, op.n_img, op.N, c.st), "softmax launch");
      return LIP_OK;
    }
    case LIP_OP_HEAD: {
      HeadP p;
      memset(&p, 0, sizeof(p));
      p.n = op.n_img; p.K = op.N; p.mode = c.head_mode; p.classifier = op.classifier; p.c = c.head_c;
      p.p = resolve(c, op.aux0); p.s = resolve(c, op.aux1);
      const lip_ref_t* in; const lip_ref_t* out;
      switch (c.head_mode) {
        case LIP_HEAD_GGN: in = &op.seg[0].a; out = &op.out; break;
        case LIP_HEAD_LT: case LIP_HEAD_OUT: in = &op.seg[0].a; out = &op.out2; break;
        case LIP_HEAD_L: case LIP_HEAD_IN: in = &op.out2; out = &op.out; break;
        default: set_error("HEAD: bad mode %d", c.head_mode); return LIP_ERR_ARG;
      }
      p.in = resolve(c, *in); p.in_ps = in->pstride;
      p.out = resolve(c, *out); p.out_ps = out->pstride;
      if (!p.in || !p.out || (p.classifier && (!p.p || !p.s))) { set_error("HEAD: bad operands (mode %d)", c.head_mode); return LIP_ERR_ARG; }
      RUN_CHECK(launch_head(p, c.P, c.st), "head launch");
      return LIP_OK;
    }
    default:
      set_error("unknown op kind %d", op.kind);
      return LIP_ERR_ARG;
  }
}

hipEvent_t next_event(lip_engine* e) {
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    e->ev_pool.push_back(ev);
  }
  return e->ev_pool[e->ev_used++];
}

int run_tape(const RunCtx& c, int which, bool skip_head) {
  const std::vector<lip_op_t>& t = c.e->tape[which];
  if (t.empty()) { set_error("tape %d is empty", which); return LIP_ERR_STATE; }
  lip_engine* me = const_cast<lip_engine*>(c.e);
  for (size_t i = 0; i < t.size(); ++i) {
    if (skip_head && t[i].kind == LIP_OP_HEAD) continue;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (me->prof) {
      e0 = next_event(me); e1 = next_event(me);
      if (e0 && e1) { me->ev_kind.push_back(t[i].kind); (void)hipEventRecord(e0, c.st); }
    }
    const int rc = run_op(c, t[i]);
    if (me->prof && e0 && e1) (void)hipEventRecord(e1, c.st);
    if (rc != LIP_OK) {
      char buf[400];
      snprintf(buf, sizeof(buf), "%s", g_err);
      set_error("tape %d op %zu (kind %d): %s", which, i, t[i].kind, buf);
      return rc;
    }
  }
  return LIP_OK;
}

// Initialise the (pc, D) output block of a summed product: alpha * V (or 0) everywhere EXCEPT the parameters a fused
// weight gradient of the backward tape will write outright (make_wgrad sets overwrite for exactly those ops).
int init_output(const RunCtx& c, int64_t D) {
  struct Range { int64_t off, len; };
  std::vector<Range> skip;
  for (const lip_op_t& op : c.e->tape[LIP_TAPE_BACKWARD]) {
    if (op.kind != LIP_OP_WGRAD) continue;
    WgradP p;
    const int rc = make_wgrad(c, op, p);
    if (rc) return rc;
    if (p.overwrite) skip.push_back({op.out.off, (int64_t)p.M * p.N});
  }
  const float* v = c.alpha != 0.f ? c.V : nullptr;
  if (skip.empty()) {
    if (v) RUN_CHECK(launch_scale_copy(c.Y, v, c.alpha, (long long)c.P * D, c.st), "scale_copy");
    else RUN_CHECK(hipMemsetAsync(c.Y, 0, sizeof(float) * (size_t)c.P * D, c.st), "memset Y");
    return LIP_OK;
  }
  std::sort(skip.begin(), skip.end(), [](const Range& a, const Range& b) { return a.off < b.off; });
  int64_t pos = 0;
  for (const Range& r : skip) {
    if (r.off < pos) { set_error("init_output: overlapping weight-gradient outputs"); return LIP_ERR_STATE; }
    RUN_CHECK(launch_scale_copy_range(c.Y, v, c.alpha, pos, r.off - pos, c.P, D, c.st), "scale_copy_range");
    pos = r.off + r.len;
  }
  RUN_CHECK(launch_scale_copy_range(c.Y, v, c.alpha, pos, D - pos, c.P, D, c.st), "scale_copy_range");
  return LIP_OK;
}

int ready(const lip_engine* e, const char* who) {
  if (!e) { set_error("%s: null engine", who); return LIP_ERR_ARG; }
  if (!e->theta || !e->prim || !e->work || e->max_chunk <= 0) { set_error("%s: engine not bound", who); return LIP_ERR_STATE; }
  if (!e->primal_done) { set_error("%s: primal pass not run", who); return LIP_ERR_STATE; }
  return LIP_OK;
}

}  // namespace

extern "C" {

int lip_abi_version(void) { return 8; }
const char* lip_last_error(void) { return g_err; }
int lip_sizeof_op(void) { return (int)sizeof(lip_op_t); }
int lip_set_precision(int32_t mode) { if (mode != 0 && mode != 1) { set_error("lip_set_precision: mode must be 0 (f32) or 1 (bf16x3)"); return LIP_ERR_ARG; } set_precision_mode(mode); return LIP_OK; }
int lip_set_split_k(int32_t on) { set_split_k_mode(on != 0); return LIP_OK; }
int lip_set_winograd(int32_t mode) { if (mode < 0 || mode > 2) { set_error("lip_set_winograd: mode must be 0 (off), 1 (auto) or 2 (every eligible launch)"); return LIP_ERR_ARG; } set_wino_mode(mode); return LIP_OK; }
int lip_get_winograd(void) { return wino_mode(); }
int lip_get_precision(void) { return precision_mode(); }

int lip_engine_create(lip_engine_t** out, int64_t D, int32_t n_img, int32_t K) {
  if (!out || D <= 0 || n_img <= 0 || K <= 0) { set_error("lip_engine_create: bad argument"); return LIP_ERR_ARG; }
  lip_engine* e = new lip_engine();
  e->D = D; e->n_img = n_img; e->K = K;
  *out = e;
  return LIP_OK;
}

int lip_engine_destroy(lip_engine_t* e) {
  if (e) for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  delete e;
  return LIP_OK;
}

int lip_engine_set_tape(lip_engine_t* e, int32_t which, const lip_op_t* ops, int32_t nops) {
  if (!e || which < 0 || which > 2 || !ops || nops <= 0) { set_error("lip_engine_set_tape: bad argument"); return LIP_ERR_ARG; }
  e->tape[which].assign(ops, ops + nops);
  if (which == LIP_TAPE_PRIMAL) e->primal_done = false;
  return LIP_OK;
}

int lip_engine_bind(lip_engine_t* e, const float* theta, const float* consts, float* prim, float* work,
                    int64_t work_floats_per_probe, int32_t max_probes_per_chunk) {
  if (!e || !theta || !prim || !work || work_floats_per_probe <= 0 || max_probes_per_chunk <= 0) {
    set_error("lip_engine_bind: bad argument");
    return LIP_ERR_ARG;
  }
  if ((((uintptr_t)prim) & 15) || (((uintptr_t)work) & 15) || (consts && (((uintptr_t)consts) & 15))) {
    set_error("lip_engine_bind: buffers must be 16-byte aligned");
    return LIP_ERR_ARG;
  }
  // re-binding only a (grown) workspace keeps the cached primal pass valid
  if (e->theta != theta || e->consts != consts || e->prim != prim) e->primal_done = false;
  e->theta = theta; e->consts = consts; e->prim = prim; e->work = work;
  e->work_pp = work_floats_per_probe; e->max_chunk = max_probes_per_chunk;
  return LIP_OK;
}

int lip_engine_primal(lip_engine_t* e, void* stream) {
  if (!e || !e->theta || !e->prim) { set_error("lip_engine_primal: engine not bound"); return LIP_ERR_STATE; }
  RunCtx c{e, nullptr, nullptr, nullptr, 1, 0, 1.f, (hipStream_t)stream};
  const int rc = run_tape(c, LIP_TAPE_PRIMAL, false);
  if (rc == LIP_OK) e->primal_done = true;
  return rc;
}

int lip_engine_profile(lip_engine_t* e, int32_t enable) {
  if (!e) { set_error("lip_engine_profile: null engine"); return LIP_ERR_ARG; }
  e->prof = enable != 0;
  e->ev_used = 0;
  e->ev_kind.clear();
  return LIP_OK;
}

int lip_engine_profile_read(lip_engine_t* e, double* ms_by_kind, int64_t* launches_by_kind, int32_t nkinds) {
  if (!e || !ms_by_kind || !launches_by_kind || nkinds <= 0) { set_error("lip_engine_profile_read: bad argument"); return LIP_ERR_ARG; }
  for (int k = 0; k < nkinds; ++k) { ms_by_kind[k] = 0.0; launches_by_kind[k] = 0; }
  for (size_t i = 0; i < e->ev_kind.size(); ++i) {
    hipEvent_t a = e->ev_pool[2 * i], b = e->ev_pool[2 * i + 1];
    RUN_CHECK(hipEventSynchronize(b), "hipEventSynchronize");
    float ms = 0.f;
    RUN_CHECK(hipEventElapsedTime(&ms, a, b), "hipEventElapsedTime");
    const int k = e->ev_kind[i];
    if (k >= 0 && k < nkinds) { ms_by_kind[k] += ms; launches_by_kind[k] += 1; }
  }
  e->ev_used = 0;
  e->ev_kind.clear();
  return LIP_OK;
}

int lip_engine_run_op(lip_engine_t* e, const lip_op_t* op, const float* V, float* Y, float* H, int32_t P, int32_t head_mode,
                      float head_c, void* stream) {
  if (!e || !op || P <= 0 || P > e->max_chunk) { set_error("lip_engine_run_op: bad argument"); return LIP_ERR_ARG; }
  if (!e->theta || !e->prim) { set_error("lip_engine_run_op: engine not bound"); return LIP_ERR_STATE; }
  RunCtx c{e, V, Y, H, P, head_mode, head_c, (hipStream_t)stream};
  return run_op(c, *op);
}

int lip_debug_run_ops(lip_engine_t* e, int32_t which, int32_t first, int32_t count, const float* V, float* Y,
                      float* H, int32_t P, int32_t head_mode, float head_c, void* stream) {
  if (!e || which < 0 || which > 2 || P <= 0 || P > e->max_chunk) { set_error("lip_debug_run_ops: bad argument"); return LIP_ERR_ARG; }
  const std::vector<lip_op_t>& t = e->tape[which];
  if (first < 0 || count < 0 || (size_t)(first + count) > t.size()) { set_error("lip_debug_run_ops: bad op range"); return LIP_ERR_ARG; }
  RunCtx c{e, V, Y, H, P, head_mode, head_c, (hipStream_t)stream};
  for (int i = first; i < first + count; ++i) {
    const int rc = run_op(c, t[i]);
    if (rc) return rc;
  }
  return LIP_OK;
}

// Probes per pass when P exceeds the workspace: equal passes (256 probes on an 85-probe workspace run 4 x 64, not
// 85 + 85 + 85 + 1 — a one-probe pass costs 2 ms of under-filled launches, a quarter of a 64-probe pass)
static inline int balanced_chunk(int P, int max_chunk) {
  const int passes = (P + max_chunk - 1) / max_chunk;
  return (P + passes - 1) / passes;
}

int lip_ggn_vp(lip_engine_t* e, const float* V, float* Y, int32_t P, float scale, float alpha, void* stream) {
  int rc = ready(e, "lip_ggn_vp");
  if (rc) return rc;
  if (!V || !Y || P <= 0) { set_error("lip_ggn_vp: bad argument"); return LIP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  for (int c0 = 0, step = balanced_chunk(P, e->max_chunk); c0 < P; c0 += step) {
    const int pc = (P - c0) < step ? (P - c0) : step;
    const float* v = V + (int64_t)c0 * e->D;
    float* y = Y + (int64_t)c0 * e->D;
    RunCtx c{e, v, y, nullptr, pc, LIP_HEAD_GGN, scale, st};
    c.fuse = true; c.alpha = alpha;
    if ((rc = init_output(c, e->D))) return rc;
    if ((rc = run_tape(c, LIP_TAPE_TANGENT, false))) return rc;
    if ((rc = run_tape(c, LIP_TAPE_BACKWARD, true))) return rc;
  }
  return LIP_OK;
}

int lip_jvp(lip_engine_t* e, const float* V, float* U, int32_t P, int32_t head_mode, float cc, void* stream) {
  int rc = ready(e, "lip_jvp");
  if (rc) return rc;
  if (!V || !U || P <= 0 || (head_mode != LIP_HEAD_LT && head_mode != LIP_HEAD_OUT)) { set_error("lip_jvp: bad argument"); return LIP_ERR_ARG; }
  const int64_t hstride = (int64_t)e->n_img * e->K;
  for (int c0 = 0, step = balanced_chunk(P, e->max_chunk); c0 < P; c0 += step) {
    const int pc = (P - c0) < step ? (P - c0) : step;
    RunCtx c{e, V + (int64_t)c0 * e->D, nullptr, U + (int64_t)c0 * hstride, pc, head_mode, cc, (hipStream_t)stream};
    if ((rc = run_tape(c, LIP_TAPE_TANGENT, false))) return rc;
  }
  return LIP_OK;
}

int lip_vjp(lip_engine_t* e, const float* U, float* Y, int32_t P, int32_t head_mode, float cc, void* stream) {
  int rc = ready(e, "lip_vjp");
  if (rc) return rc;
  if (!U || !Y || P <= 0 || (head_mode != LIP_HEAD_L && head_mode != LIP_HEAD_IN)) { set_error("lip_vjp: bad argument"); return LIP_ERR_ARG; }
  const int64_t hstride = (int64_t)e->n_img * e->K;
  hipStream_t st = (hipStream_t)stream;
  for (int c0 = 0, step = balanced_chunk(P, e->max_chunk); c0 < P; c0 += step) {
    const int pc = (P - c0) < step ? (P - c0) : step;
    float* y = Y + (int64_t)c0 * e->D;
    RunCtx c{e, nullptr, y, const_cast<float*>(U) + (int64_t)c0 * hstride, pc, head_mode, cc, st};
    c.fuse = true; c.alpha = 0.f;
    if ((rc = init_output(c, e->D))) return rc;
    if ((rc = run_tape(c, LIP_TAPE_BACKWARD, false))) return rc;
  }
  return LIP_OK;
}

int lip_vjp_rows(lip_engine_t* e, const float* U, float* Y, int32_t P, int32_t head_mode, float cc, void* stream) {
  int rc = ready(e, "lip_vjp_rows");
  if (rc) return rc;
  if (!U || !Y || P <= 0 || (head_mode != LIP_HEAD_L && head_mode != LIP_HEAD_IN)) { set_error("lip_vjp_rows: bad argument"); return LIP_ERR_ARG; }
  const int64_t hstride = (int64_t)e->n_img * e->K, ystride = (int64_t)e->n_img * e->D;
  hipStream_t st = (hipStream_t)stream;
  for (int c0 = 0, step = balanced_chunk(P, e->max_chunk); c0 < P; c0 += step) {
    const int pc = (P - c0) < step ? (P - c0) : step;
    float* y = Y + (int64_t)c0 * ystride;
    RUN_CHECK(hipMemsetAsync(y, 0, sizeof(float) * (size_t)pc * ystride, st), "memset Y");
    RunCtx c{e, nullptr, y, const_cast<float*>(U) + (int64_t)c0 * hstride, pc, head_mode, cc, st, true};
    if ((rc = run_tape(c, LIP_TAPE_BACKWARD, false))) return rc;
  }
  return LIP_OK;
}

}  // extern "C"
