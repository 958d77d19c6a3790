// program.hip -- C ABI front-end: direct conv entry point and the "program"
// recorder/executor (a native launch list with slot-relative pointers, hipEvent
// per-op timing and hipGraph capture/replay).
#include <stdlib.h>
#include <string.h>

#include <string>
#include <atomic>
#include <vector>

#include "egn_internal.h"

extern "C" int egn_conv_config_kind(int cfg);

extern "C" int egn_version(void) { return (1 << 16) | 0; }

extern "C" const char* egn_strerror(int code) {
  if (code == 0) return "ok";
  if (code == EGN_E_BADARG) return "egonet_hip: bad argument / unsupported shape";
  if (code == EGN_E_LDS) return "egonet_hip: tile does not fit in LDS";
  if (code == EGN_E_STATE) return "egonet_hip: program in wrong state";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "egonet_hip: unknown error";
}

static int fill_conv_args(ConvArgs& a, const float* x, const float* wpack, const float* scale,
                          const float* shift, const float* res, float* y, int N, int H, int W, int Cin,
                          int cs_in, int Cout, int cs_out, int KH, int KW, int stride, int pad, int act,
                          int out_nchw) {
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = wpack; a.scale = scale; a.shift = shift; a.res = res; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.cs_in = cs_in;
  a.Cout = Cout; a.cs_out = cs_out;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  a.act = act; a.out_nchw = out_nchw;
  if (out_nchw && res) return EGN_E_BADARG;
  return 0;
}

// convolution-class launches issued through the direct entry points (egn_conv2d_f32,
// egn_conv2d_bnstats_f32, egn_conv2d_wgrad_f32) since the library was loaded: the training tape and the
// autograd bridge do not go through programs, this is their "ran on this library's kernels" proof
std::atomic<long> g_egn_direct_convs{0};
extern "C" long egn_direct_conv_count(void) { return g_egn_direct_convs.load(std::memory_order_relaxed); }

extern "C" int egn_conv2d_f32(const float* x, const float* wpack, const float* scale, const float* shift,
                              const float* res, float* y, int N, int H, int W, int Cin, int cs_in, int Cout,
                              int cs_out, int KH, int KW, int stride, int pad, int act, int out_nchw,
                              int cfg, void* stream) {
  g_egn_direct_convs.fetch_add(1, std::memory_order_relaxed);
  ConvArgs a;
  int rc = fill_conv_args(a, x, wpack, scale, shift, res, y, N, H, W, Cin, cs_in, Cout, cs_out, KH, KW,
                          stride, pad, act, out_nchw);
  if (rc) return rc;
  size_t lds;
  rc = egn_conv_plan(a, cfg, lds);
  if (rc) return rc;
  return egn_conv_launch(a, cfg, (hipStream_t)stream);
}

// raw convolution (scale 1, shift 0, no residual, no activation) whose epilogue also writes per-tile
// partial column sums / sums of squares of its output: BatchNorm's batch statistics without a second
// pass over z.  Only tile configurations with fused statistics (egn_conv2d_bnstats_rows > 0).
extern "C" long egn_conv2d_bnstats_rows(int N, int H, int W, int Cin, int cs_in, int Cout, int cs_out, int KH,
                                        int KW, int stride, int pad, int cfg) {
  ConvArgs a;
  if (fill_conv_args(a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, cs_in, Cout, cs_out, KH,
                     KW, stride, pad, 0, 0))
    return 0;
  size_t lds;
  if (cfg < 1 || egn_conv_plan(a, cfg, lds)) return 0;
  return egn_conv_stats_rows(a, cfg);
}

extern "C" int egn_conv2d_bnstats_f32(const float* x, const float* wpack, const float* ones, const float* zeros,
                                      float* y, int N, int H, int W, int Cin, int cs_in, int Cout, int cs_out,
                                      int KH, int KW, int stride, int pad, int cfg, double* partials,
                                      long partial_rows, void* stream) {
  g_egn_direct_convs.fetch_add(1, std::memory_order_relaxed);
  ConvArgs a;
  int rc = fill_conv_args(a, x, wpack, ones, zeros, nullptr, y, N, H, W, Cin, cs_in, Cout, cs_out, KH, KW, stride,
                          pad, EGN_ACT_NONE, 0);
  if (rc) return rc;
  if (cfg < 1 || !partials) return EGN_E_BADARG;
  size_t lds;
  rc = egn_conv_plan(a, cfg, lds);
  if (rc) return rc;
  const int rows = egn_conv_stats_rows(a, cfg);
  if (rows <= 0 || partial_rows < rows) return EGN_E_BADARG;
  a.stats = partials;
  return egn_conv_launch(a, cfg, (hipStream_t)stream);
}

// egn_conv2d_f32 with the two optional side tables of the training tape [round 5]:
//   partials / partial_rows: BatchNorm partial statistics of the stored output, as egn_conv2d_bnstats_f32 (NULL: none;
//                            else egn_conv2d_bnstats_rows(cfg) must be > 0 and <= partial_rows);
//   tickets / ticket_words:  zeroed words for the K-split configurations (egn_conv2d_ticket_words(cfg) of them): the
//                            layer then runs as ONE kernel as inside a program and leaves the words zero; launches that
//                            share the words must be ordered (one stream).  NULL: the three-launch form.
extern "C" long egn_conv2d_ticket_words(int N, int H, int W, int Cin, int cs_in, int Cout, int cs_out, int KH, int KW,
                                        int stride, int pad, int cfg) {
  ConvArgs a;
  if (fill_conv_args(a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, cs_in, Cout, cs_out, KH,
                     KW, stride, pad, 0, 0))
    return 0;
  size_t lds;
  if (cfg < 1 || egn_conv_plan(a, cfg, lds)) return 0;
  return egn_conv_ticket_count(a, cfg);
}

extern "C" int egn_conv2d_ex_f32(const float* x, const float* wpack, const float* scale, const float* shift,
                                 const float* res, float* y, int N, int H, int W, int Cin, int cs_in, int Cout,
                                 int cs_out, int KH, int KW, int stride, int pad, int act, int cfg, double* partials,
                                 long partial_rows, unsigned* tickets, long ticket_words, void* stream) {
  g_egn_direct_convs.fetch_add(1, std::memory_order_relaxed);
  ConvArgs a;
  int rc = fill_conv_args(a, x, wpack, scale, shift, res, y, N, H, W, Cin, cs_in, Cout, cs_out, KH, KW, stride, pad,
                          act, 0);
  if (rc) return rc;
  if (cfg < 1) return EGN_E_BADARG;
  size_t lds;
  rc = egn_conv_plan(a, cfg, lds);
  if (rc) return rc;
  if (partials) {
    const int rows = egn_conv_stats_rows(a, cfg);
    if (rows <= 0 || partial_rows < rows) return EGN_E_BADARG;
    a.stats = partials;
  }
  if (tickets) {
    const int ntk = egn_conv_ticket_count(a, cfg);
    if (ntk > 0) {
      if (ticket_words < ntk) return EGN_E_BADARG;
      a.tickets = tickets;
    }
  }
  return egn_conv_launch(a, cfg, (hipStream_t)stream);
}

extern "C" int egn_conv_plan_query(int N, int H, int W, int Cin, int cs_in, int Cout, int cs_out, int KH,
                                   int KW, int stride, int pad, int out_nchw, int cfg, int* out) {
  if (!out) return EGN_E_BADARG;
  ConvArgs a;
  int rc = fill_conv_args(a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, cs_in, Cout,
                          cs_out, KH, KW, stride, pad, 0, out_nchw);
  if (rc) return rc;
  size_t lds;
  rc = egn_conv_plan(a, cfg, lds);
  if (rc) return rc;
  int tm, tn;
  egn_conv_config_info(cfg, &tm, &tn);
  const ConvConfig* cf = egn_conv_config(cfg);
  out[0] = cfg; out[1] = cf->wm; out[2] = cf->wn; out[3] = cf->mt; out[4] = cf->nt;
  out[5] = a.TH; out[6] = a.TW; out[7] = a.TNB; out[8] = a.tps; out[9] = (int)lds;
  out[10] = a.tiles_x * a.tiles_y * ((a.N + a.TNB - 1) / a.TNB);
  out[11] = (a.CoutP + tn - 1) / tn;
  if (egn_conv_config_kind(cfg) == 1 && egn_wino_cot(Cout)) out[11] = Cout / egn_wino_cot(Cout);
  return 0;
}

// ---------------------------------------------------------------------------
// programs
// ---------------------------------------------------------------------------
enum OpKind { OP_CONV = 1, OP_FUSE = 2, OP_NCHW2NHWC = 3, OP_NHWC2NCHW = 4, OP_RAMPS = 5, OP_DECODE = 6,
              OP_FORK = 7, OP_JOIN = 8, OP_PIXSHUF = 9, OP_PWPAIR = 10 };
constexpr int kMaxLanes = 4;  // concurrent launch lanes (HRNet has at most 4 branches)

struct Op {
  int kind;
  int lane = 0;  // launch lane (0 = the caller's stream, 1.. = side streams) inside a fork/join region
  // conv
  ConvArgs conv;
  int cfg;
  egn_ref r[8];  // pointer refs, meaning depends on kind
  // generic ints
  int i[12];
  std::string tag;
  double flops, bytes;
};

struct egn_program {
  std::vector<void*> slots;
  std::vector<Op> ops;
  std::vector<void*> owned;   // device words of the program's own (ConvArgs::tickets of K-split convs): freed with it
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int cur_lane = 0;
  // side streams / events for fork-join regions (created on first use)
  hipStream_t side[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr;
  hipEvent_t ev_join[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  // "a program never runs beside itself" (one arena, one set of side streams, one set of ticket words) is enforced, not
  // assumed [round 5]: every eager run / replay records ev_done on its stream, and a run on ANOTHER stream waits for it
  hipEvent_t ev_done = nullptr;
  hipStream_t last_stream = nullptr;
  bool ran = false;
  // K-split ops: word 0 of an op's ticket buffer is raised by a block whose bounded wait for its partner ran out
  // (csrc/conv_wino4.hip).  Behind every run one tiny kernel ORs the ops' words into err_dev (and clears them), one
  // 4-byte copy brings it to the pinned err_host; the NEXT run / replay looks at it: a raised word fails that call with
  // EGN_E_STATE after all ticket words were zeroed again [round 6, ADVICE r5: nothing read the word before].
  std::vector<unsigned*> tickets;      // the ticket buffers of the K-split ops (also in `owned`)
  std::vector<size_t> ticket_words;
  unsigned** err_ptrs = nullptr;       // device array of the buffers' addresses (built at the first run)
  unsigned* err_dev = nullptr;
  unsigned* err_host = nullptr;        // pinned
  hipStream_t init_stream = nullptr;   // zeroing of new ticket words (never the legacy stream: no device-wide sync)
};

__global__ void egn_ticket_errors_kernel(unsigned** bufs, int n, unsigned* any) {
  unsigned e = 0;
  for (int k = threadIdx.x; k < n; k += 64)
    if (bufs[k][0]) { e = 1u; bufs[k][0] = 0u; }
  if (e) *any = 1u;
}

// behind the ops of a run (eager, or recorded into the capture): gather the error words, copy to the pinned mirror
static int issue_ticket_check(egn_program* p, hipStream_t s) {
  if (p->tickets.empty()) return 0;
  if (!p->err_ptrs) {
    EGN_CHECK_HIP(hipMalloc(&p->err_ptrs, p->tickets.size() * sizeof(unsigned*)));
    EGN_CHECK_HIP(hipMalloc(&p->err_dev, sizeof(unsigned)));
    EGN_CHECK_HIP(hipHostMalloc(&p->err_host, sizeof(unsigned), hipHostMallocDefault));
    *p->err_host = 0u;
    // (blocking copies from pageable memory: the program is being run for the first time, nothing of it is in flight)
    EGN_CHECK_HIP(hipMemcpy(p->err_ptrs, p->tickets.data(), p->tickets.size() * sizeof(unsigned*), hipMemcpyHostToDevice));
    EGN_CHECK_HIP(hipMemset(p->err_dev, 0, sizeof(unsigned)));
  }
  hipLaunchKernelGGL(egn_ticket_errors_kernel, dim3(1), dim3(64), 0, s, p->err_ptrs, (int)p->tickets.size(), p->err_dev);
  EGN_CHECK_HIP(hipGetLastError());
  EGN_CHECK_HIP(hipMemcpyAsync(p->err_host, p->err_dev, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  return 0;
}
// at the top of a run / replay: did an EARLIER run raise an error word?  (its copy has landed if ev_done has)
static int consume_ticket_error(egn_program* p, hipStream_t s) {
  if (!p->err_host || !p->ran || hipEventQuery(p->ev_done) != hipSuccess || !*p->err_host) return 0;
  *p->err_host = 0u;
  EGN_CHECK_HIP(hipMemsetAsync(p->err_dev, 0, sizeof(unsigned), s));
  for (size_t k = 0; k < p->tickets.size(); ++k)
    EGN_CHECK_HIP(hipMemsetAsync(p->tickets[k], 0, p->ticket_words[k] * sizeof(unsigned), s));
  return EGN_E_STATE;
}

// order this run behind the previous one of the same program if that was issued on another stream (same stream: already
// ordered).  Not while the stream is being captured (the graph's own edges order a replay; replays are ordered below).
static int order_behind_last_run(egn_program* p, hipStream_t s, bool* capturing) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  *capturing = hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  if (*capturing) return 0;
  if (!p->ev_done) EGN_CHECK_HIP(hipEventCreateWithFlags(&p->ev_done, hipEventDisableTiming));
  if (p->ran && p->last_stream != s) EGN_CHECK_HIP(hipStreamWaitEvent(s, p->ev_done, 0));
  return 0;
}
static int mark_run_issued(egn_program* p, hipStream_t s, bool capturing) {
  if (capturing) return 0;
  EGN_CHECK_HIP(hipEventRecord(p->ev_done, s));
  p->last_stream = s;
  p->ran = true;
  return 0;
}

static inline void* resolve(const egn_program* p, const egn_ref& r) {
  if (r.slot < 0) return nullptr;
  return (char*)p->slots[r.slot] + r.off;
}
static inline bool ref_ok(const egn_program* p, const egn_ref& r) {
  return r.slot < (int)p->slots.size();
}

extern "C" egn_program* egn_program_create(int nslots) {
  if (nslots < 1 || nslots > 64) return nullptr;
  egn_program* p = new egn_program();
  p->slots.assign(nslots, nullptr);
  return p;
}

// A captured graph (and the side streams / ticket words of a program) must not be released while a run or replay of
// the program is still executing: wait for the event its last run recorded.  (Before round 5 programs were only captured
// by tools and tests that synchronised themselves; the engine now replays small batches as graphs and drops programs
// whenever the weights change -- right behind an asynchronous replay.)
static void wait_for_last_run(egn_program* p) {
  if (p->ev_done && p->ran) hipEventSynchronize(p->ev_done);
}

extern "C" void egn_program_destroy(egn_program* p) {
  if (!p) return;
  wait_for_last_run(p);
  if (p->exec) hipGraphExecDestroy(p->exec);
  if (p->graph) hipGraphDestroy(p->graph);
  for (void* d : p->owned) hipFree(d);
  if (p->err_ptrs) hipFree(p->err_ptrs);
  if (p->err_dev) hipFree(p->err_dev);
  if (p->err_host) hipHostFree(p->err_host);
  if (p->init_stream) hipStreamDestroy(p->init_stream);
  for (int k = 1; k < kMaxLanes; ++k) {
    if (p->side[k]) hipStreamDestroy(p->side[k]);
    if (p->ev_join[k]) hipEventDestroy(p->ev_join[k]);
  }
  if (p->ev_fork) hipEventDestroy(p->ev_fork);
  if (p->ev_done) hipEventDestroy(p->ev_done);
  delete p;
}

extern "C" int egn_program_bind(egn_program* p, int slot, void* base) {
  if (!p || slot < 0 || slot >= (int)p->slots.size()) return EGN_E_BADARG;
  if (p->slots[slot] != base && p->exec) {  // a captured graph holds the old addresses
    wait_for_last_run(p);
    hipGraphExecDestroy(p->exec); p->exec = nullptr;
    hipGraphDestroy(p->graph); p->graph = nullptr;
  }
  p->slots[slot] = base;
  return 0;
}

extern "C" int egn_program_num_ops(const egn_program* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int egn_program_add_conv2d(egn_program* p, egn_ref x, egn_ref wpack, egn_ref scale, egn_ref shift,
                                      egn_ref res, egn_ref y, int N, int H, int W, int Cin, int cs_in,
                                      int Cout, int cs_out, int KH, int KW, int stride, int pad, int act,
                                      int out_nchw, int cfg) {
  if (!p) return EGN_E_BADARG;
  Op op;
  op.kind = OP_CONV;
  op.flops = op.bytes = 0;
  int rc = fill_conv_args(op.conv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, cs_in,
                          Cout, cs_out, KH, KW, stride, pad, act, out_nchw);
  if (rc) return rc;
  if (out_nchw && res.slot >= 0) return EGN_E_BADARG;
  size_t lds;
  rc = egn_conv_plan(op.conv, cfg, lds);
  if (rc) return rc;
  op.cfg = cfg;
  op.r[0] = x; op.r[1] = wpack; op.r[2] = scale; op.r[3] = shift; op.r[4] = res; op.r[5] = y;
  for (int k = 0; k < 6; ++k)
    if (!ref_ok(p, op.r[k])) return EGN_E_BADARG;
  op.flops = 2.0 * N * op.conv.Ho * op.conv.Wo * (double)Cout * Cin * KH * KW;
  op.lane = p->cur_lane;
  // K-split configurations (conv_wino4.hip) run as one kernel when the op brings a ticket word per item pair: they
  // belong to the op (a program never runs beside itself), are zeroed here and left zero by every launch
  // (EGONET_AMD_NO_TICKETS=1: no words -- the three-launch form of such a layer, for A/B runs and as a fallback)
  static const bool no_tickets = [] { const char* e = getenv("EGONET_AMD_NO_TICKETS"); return e && e[0] == '1'; }();
  const int ntk = no_tickets ? 0 : egn_conv_ticket_count(op.conv, cfg);
  if (ntk > 0) {
    void* d = nullptr;
    EGN_CHECK_HIP(hipMalloc(&d, (size_t)ntk * sizeof(unsigned)));
    p->owned.push_back(d);
    // (programs run on the caller's non-blocking streams, which do not order themselves behind the legacy stream: the
    // words are zero in memory before this call returns -- a non-zero ticket would make both halves of a pair wait)
    // [round 6, ADVICE r5] on a private stream + a wait for THAT stream: no device-wide synchronisation per K-split op
    // (the tuner builds one-op programs by the hundred) -- programs are built outside stream captures
    if (!p->init_stream) EGN_CHECK_HIP(hipStreamCreateWithFlags(&p->init_stream, hipStreamNonBlocking));
    EGN_CHECK_HIP(hipMemsetAsync(d, 0, (size_t)ntk * sizeof(unsigned), p->init_stream));
    EGN_CHECK_HIP(hipStreamSynchronize(p->init_stream));
    op.conv.tickets = static_cast<unsigned*>(d);
    p->tickets.push_back(static_cast<unsigned*>(d));
    p->ticket_words.push_back((size_t)ntk);
    if (p->err_ptrs) { hipFree(p->err_ptrs); p->err_ptrs = nullptr; hipFree(p->err_dev); p->err_dev = nullptr;
                       hipHostFree(p->err_host); p->err_host = nullptr; }      // (an op added after a run: rebuilt)
  }
  p->ops.push_back(op);
  return 0;
}

extern "C" int egn_program_add_fuse(egn_program* p, egn_ref y, int N, int H, int W, int C, int cs, int nterms,
                                    const egn_ref* terms, const int* shifts, int relu) {
  if (!p || nterms < 1 || nterms > 4 || cs % 4 || C > cs) return EGN_E_BADARG;
  Op op;
  op.kind = OP_FUSE;
  op.flops = op.bytes = 0;
  op.r[0] = y;
  for (int k = 0; k < nterms; ++k) {
    op.r[1 + k] = terms[k];
    op.i[6 + k] = shifts[k];
    if (!ref_ok(p, terms[k])) return EGN_E_BADARG;
  }
  op.i[0] = N; op.i[1] = H; op.i[2] = W; op.i[3] = C; op.i[4] = cs; op.i[5] = nterms; op.i[10] = relu;
  op.lane = p->cur_lane;
  p->ops.push_back(op);
  return 0;
}

static int add_simple(egn_program* p, int kind, egn_ref a, egn_ref b, int i0, int i1, int i2, int i3, int i4,
                      int i5) {
  if (!p || !ref_ok(p, a) || !ref_ok(p, b)) return EGN_E_BADARG;
  Op op;
  op.kind = kind;
  op.flops = op.bytes = 0;
  op.r[0] = a; op.r[1] = b;
  op.i[0] = i0; op.i[1] = i1; op.i[2] = i2; op.i[3] = i3; op.i[4] = i4; op.i[5] = i5;
  op.lane = p->cur_lane;
  p->ops.push_back(op);
  return 0;
}

extern "C" int egn_program_add_nchw_to_nhwc(egn_program* p, egn_ref x, egn_ref y, int N, int C, int H, int W,
                                            int cs) {
  if (cs % 4 || cs < C) return EGN_E_BADARG;
  return add_simple(p, OP_NCHW2NHWC, x, y, N, C, H, W, cs, 0);
}
extern "C" int egn_program_add_nhwc_to_nchw(egn_program* p, egn_ref x, egn_ref y, int N, int C, int H, int W,
                                            int cs) {
  if (cs < C) return EGN_E_BADARG;
  return add_simple(p, OP_NHWC2NCHW, x, y, N, C, H, W, cs, 0);
}
extern "C" int egn_program_add_pixel_shuffle(egn_program* p, egn_ref x, egn_ref y, int N, int C, int H, int W,
                                             int cs, int up) {
  if (up < 1 || cs < C * up * up) return EGN_E_BADARG;
  return add_simple(p, OP_PIXSHUF, x, y, N, C, H, W, cs, up);
}
extern "C" int egn_program_add_ramps(egn_program* p, egn_ref y, int N, int H, int W, int cs, int c0) {
  egn_ref none = {-1, 0};
  return add_simple(p, OP_RAMPS, y, none, N, H, W, cs, c0, 0);
}
extern "C" int egn_program_add_decode(egn_program* p, egn_ref hm, int N, int K, int H, int W, int mode,
                                      egn_ref out_xy, egn_ref out_max, egn_ref out_idx) {
  if (!p || !ref_ok(p, hm) || !ref_ok(p, out_xy) || !ref_ok(p, out_max) || !ref_ok(p, out_idx))
    return EGN_E_BADARG;
  Op op;
  op.kind = OP_DECODE;
  op.flops = op.bytes = 0;
  op.r[0] = hm; op.r[1] = out_xy; op.r[2] = out_max; op.r[3] = out_idx;
  op.i[0] = N; op.i[1] = K; op.i[2] = H; op.i[3] = W; op.i[4] = mode;
  op.lane = p->cur_lane;
  p->ops.push_back(op);
  return 0;
}

// layer1's 1x1 pair (conv_pw.hip, egn_pw_pair_f32): out = act(h . w3^T + shift3 (+ res)), hn = relu(out . w1^T + shift1);
// w1 / shift1 / hn with slot < 0: the first product alone
extern "C" int egn_program_add_pw_pair(egn_program* p, egn_ref h, egn_ref res, egn_ref w3, egn_ref shift3, egn_ref w1,
                                       egn_ref shift1, egn_ref out, egn_ref hn, int M, int relu1) {
  if (!p || M <= 0 || M % 32 || h.slot < 0 || w3.slot < 0 || shift3.slot < 0 || out.slot < 0) return EGN_E_BADARG;
  if ((w1.slot < 0) != (hn.slot < 0) || (w1.slot >= 0 && shift1.slot < 0)) return EGN_E_BADARG;
  Op op;
  op.kind = OP_PWPAIR;
  op.r[0] = h; op.r[1] = res; op.r[2] = w3; op.r[3] = shift3; op.r[4] = w1; op.r[5] = shift1; op.r[6] = out; op.r[7] = hn;
  for (int k = 0; k < 8; ++k)
    if (!ref_ok(p, op.r[k])) return EGN_E_BADARG;
  op.i[0] = M; op.i[1] = relu1;
  op.flops = 2.0 * M * 64.0 * 256.0 * (w1.slot >= 0 ? 2.0 : 1.0);
  op.bytes = 4.0 * M * (64.0 + 256.0 * (res.slot >= 0 ? 2.0 : 1.0) + (w1.slot >= 0 ? 64.0 : 0.0));
  op.lane = p->cur_lane;
  p->ops.push_back(op);
  return 0;
}

extern "C" int egn_program_fork(egn_program* p) {
  if (!p) return EGN_E_BADARG;
  Op op;
  op.kind = OP_FORK;
  op.flops = op.bytes = 0;
  p->cur_lane = 0;
  p->ops.push_back(op);
  return 0;
}
extern "C" int egn_program_join(egn_program* p) {
  if (!p) return EGN_E_BADARG;
  Op op;
  op.kind = OP_JOIN;
  op.flops = op.bytes = 0;
  p->cur_lane = 0;
  p->ops.push_back(op);
  return 0;
}
extern "C" int egn_program_set_lane(egn_program* p, int lane) {
  if (!p || lane < 0 || lane >= kMaxLanes) return EGN_E_BADARG;
  p->cur_lane = lane;
  return 0;
}

extern "C" int egn_program_tag(egn_program* p, const char* tag, double flops, double bytes) {
  if (!p || p->ops.empty()) return EGN_E_STATE;
  Op& op = p->ops.back();
  op.tag = tag ? tag : "";
  if (flops > 0) op.flops = flops;
  op.bytes = bytes;
  return 0;
}

extern "C" int egn_program_op_info(const egn_program* p, int i, int* kind, double* flops, double* bytes,
                                   char* tag, int tag_len) {
  if (!p || i < 0 || i >= (int)p->ops.size()) return EGN_E_BADARG;
  const Op& op = p->ops[i];
  if (kind) *kind = op.kind;
  if (flops) *flops = op.flops;
  if (bytes) *bytes = op.bytes;
  if (tag && tag_len > 0) {
    strncpy(tag, op.tag.c_str(), tag_len - 1);
    tag[tag_len - 1] = 0;
  }
  return 0;
}

// kernel launches issued through programs since the library was loaded (egn_launch_count): lets a
// caller PROVE that a forward ran on this library's kernels and not on some other route
static std::atomic<long> g_launches{0};
static thread_local bool t_recording = false;   // this thread is inside egn_program_capture
extern "C" long egn_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

static int launch_op(egn_program* p, Op& op, hipStream_t s) {
  if (!t_recording) g_launches.fetch_add(1, std::memory_order_relaxed);
  switch (op.kind) {
    case OP_CONV: {
      ConvArgs a = op.conv;
      a.x = (const float*)resolve(p, op.r[0]);
      a.w = (const float*)resolve(p, op.r[1]);
      a.scale = (const float*)resolve(p, op.r[2]);
      a.shift = (const float*)resolve(p, op.r[3]);
      a.res = (const float*)resolve(p, op.r[4]);
      a.y = (float*)resolve(p, op.r[5]);
      return egn_conv_launch(a, op.cfg, s);
    }
    case OP_FUSE: {
      const float* terms[4];
      int shifts[4];
      for (int k = 0; k < op.i[5]; ++k) {
        terms[k] = (const float*)resolve(p, op.r[1 + k]);
        shifts[k] = op.i[6 + k];
      }
      return egn_fuse_sum_relu_f32((float*)resolve(p, op.r[0]), op.i[0], op.i[1], op.i[2], op.i[3], op.i[4],
                                   op.i[5], terms, shifts, op.i[10], s);
    }
    case OP_NCHW2NHWC:
      return egn_nchw_to_nhwc_f32((const float*)resolve(p, op.r[0]), (float*)resolve(p, op.r[1]), op.i[0],
                                  op.i[1], op.i[2], op.i[3], op.i[4], s);
    case OP_NHWC2NCHW:
      return egn_nhwc_to_nchw_f32((const float*)resolve(p, op.r[0]), (float*)resolve(p, op.r[1]), op.i[0],
                                  op.i[1], op.i[2], op.i[3], op.i[4], s);
    case OP_PIXSHUF:
      return egn_pixel_shuffle_nhwc_to_nchw_f32((const float*)resolve(p, op.r[0]), (float*)resolve(p, op.r[1]),
                                                op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], s);
    case OP_RAMPS:
      return egn_fill_coord_ramps_f32((float*)resolve(p, op.r[0]), op.i[0], op.i[1], op.i[2], op.i[3],
                                      op.i[4], s);
    case OP_PWPAIR:
      return egn_pw_pair_f32((const float*)resolve(p, op.r[0]), (const float*)resolve(p, op.r[1]),
                             (const float*)resolve(p, op.r[2]), (const float*)resolve(p, op.r[3]),
                             (const float*)resolve(p, op.r[4]), (const float*)resolve(p, op.r[5]),
                             (float*)resolve(p, op.r[6]), (float*)resolve(p, op.r[7]), op.i[0], op.i[1], s);
    case OP_DECODE:
      return egn_decode_heatmaps_f32((const float*)resolve(p, op.r[0]), op.i[0], op.i[1], op.i[2], op.i[3],
                                     op.i[4], (float*)resolve(p, op.r[1]), (float*)resolve(p, op.r[2]),
                                     (int32_t*)resolve(p, op.r[3]), s);
  }
  return EGN_E_BADARG;
}

static int ensure_lanes(egn_program* p) {
  if (p->ev_fork) return 0;
  EGN_CHECK_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
  for (int k = 1; k < kMaxLanes; ++k) {
    EGN_CHECK_HIP(hipStreamCreateWithFlags(&p->side[k], hipStreamNonBlocking));
    EGN_CHECK_HIP(hipEventCreateWithFlags(&p->ev_join[k], hipEventDisableTiming));
  }
  return 0;
}

// Launch every op.  Ops between FORK and JOIN go to their lane's stream: lane 0
// is the caller's stream, lanes 1.. are side streams that wait for the fork
// point and are waited for at the join (the HRNet branches / fuse outputs are
// independent, so their kernels fill each other's prologue, epilogue and tail
// gaps).  Works unchanged under stream capture (the events become graph edges).
extern "C" int egn_program_run(egn_program* p, void* stream) {
  if (!p) return EGN_E_BADARG;
  hipStream_t main = (hipStream_t)stream;
  bool used[kMaxLanes] = {false, false, false, false};
  bool capturing = false;
  int rc_o = order_behind_last_run(p, main, &capturing);
  if (rc_o) return rc_o;
  if (!capturing) {
    rc_o = consume_ticket_error(p, main);      // an earlier run's K-split wait ran out: that run's output was invalid
    if (rc_o) return rc_o;
  }
  for (Op& op : p->ops) {
    if (op.kind == OP_FORK) {
      int rc = ensure_lanes(p);
      if (rc) return rc;
      EGN_CHECK_HIP(hipEventRecord(p->ev_fork, main));
      for (int k = 1; k < kMaxLanes; ++k) used[k] = false;
      continue;
    }
    if (op.kind == OP_JOIN) {
      for (int k = 1; k < kMaxLanes; ++k)
        if (used[k]) {
          EGN_CHECK_HIP(hipEventRecord(p->ev_join[k], p->side[k]));
          EGN_CHECK_HIP(hipStreamWaitEvent(main, p->ev_join[k], 0));
          used[k] = false;
        }
      continue;
    }
    hipStream_t s = main;
    if (op.lane > 0 && p->ev_fork) {
      if (!used[op.lane]) {
        EGN_CHECK_HIP(hipStreamWaitEvent(p->side[op.lane], p->ev_fork, 0));
        used[op.lane] = true;
      }
      s = p->side[op.lane];
    }
    int rc = launch_op(p, op, s);
    if (rc) return rc;
  }
  if (!capturing || p->err_ptrs) {             // (a capture records the check only if its buffers exist already)
    int rc = issue_ticket_check(p, main);
    if (rc) return rc;
  }
  return mark_run_issued(p, main, capturing);
}

extern "C" int egn_program_run_timed(egn_program* p, void* stream, float* ms, int n_ms) {
  if (!p || !ms || n_ms < (int)p->ops.size()) return EGN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = p->ops.size();
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev) EGN_CHECK_HIP(hipEventCreate(&e));
  // [round 6, ADVICE r5] ordered behind the program's previous run like every other run (one arena, one set of words)
  bool capturing = false;
  int rc = order_behind_last_run(p, s, &capturing);
  if (!rc && capturing) rc = EGN_E_STATE;      // (a timed run synchronises: never inside a capture)
  if (!rc) rc = consume_ticket_error(p, s);
  if (rc) { for (auto& e : ev) hipEventDestroy(e); return rc; }
  EGN_CHECK_HIP(hipEventRecord(ev[0], s));
  for (size_t i = 0; i < n && !rc; ++i) {  // serial on the caller's stream: clean per-kernel times
    if (p->ops[i].kind != OP_FORK && p->ops[i].kind != OP_JOIN) rc = launch_op(p, p->ops[i], s);
    if (!rc) rc = (int)hipEventRecord(ev[i + 1], s);
  }
  if (!rc) rc = issue_ticket_check(p, s);
  if (!rc) rc = mark_run_issued(p, s, false);
  if (!rc) rc = (int)hipStreamSynchronize(s);
  if (!rc)
    for (size_t i = 0; i < n; ++i) {
      float t = 0.f;
      hipEventElapsedTime(&t, ev[i], ev[i + 1]);
      ms[i] = t;
    }
  for (auto& e : ev) hipEventDestroy(e);
  return rc;
}

extern "C" int egn_program_ticket_ops(const egn_program* p) { return p ? (int)p->tickets.size() : 0; }
extern "C" int egn_program_poke_ticket(egn_program* p, int op, int word, unsigned value) {
  if (!p || op < 0 || op >= (int)p->tickets.size() || word < 0 || (size_t)word >= p->ticket_words[op]) return EGN_E_BADARG;
  wait_for_last_run(p);
  EGN_CHECK_HIP(hipMemcpy(p->tickets[op] + word, &value, sizeof(unsigned), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int egn_program_capture(egn_program* p, void* stream) {
  if (!p) return EGN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (p->exec) { wait_for_last_run(p); hipGraphExecDestroy(p->exec); p->exec = nullptr; }
  if (p->graph) { hipGraphDestroy(p->graph); p->graph = nullptr; }
  for (const Op& op : p->ops)  // side streams / events must exist before the capture starts
    if (op.kind == OP_FORK) {
      int rc0 = ensure_lanes(p);
      if (rc0) return rc0;
      break;
    }
  EGN_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  t_recording = true;      // (recorded, not executed: egn_launch_count counts launches that ran -- replays count themselves)
  int rc = egn_program_run(p, stream);
  t_recording = false;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(s, &g);
  if (rc) { if (g) hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) return (int)e;
  p->graph = g;
  EGN_CHECK_HIP(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
  return 0;
}

extern "C" int egn_program_replay(egn_program* p, void* stream) {
  if (!p || !p->exec) return EGN_E_STATE;
  long nk = 0;
  for (const Op& op : p->ops) nk += (op.kind != OP_FORK && op.kind != OP_JOIN);
  g_launches.fetch_add(nk, std::memory_order_relaxed);
  bool capturing = false;
  int rc = order_behind_last_run(p, (hipStream_t)stream, &capturing);
  if (rc) return rc;
  if (!capturing) {
    rc = consume_ticket_error(p, (hipStream_t)stream);
    if (rc) return rc;
  }
  EGN_CHECK_HIP(hipGraphLaunch(p->exec, (hipStream_t)stream));
  return mark_run_issued(p, (hipStream_t)stream, capturing);
}
