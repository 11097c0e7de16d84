// crx_api.hip -- the C ABI of libcrx (include/crx.h) on top of the gfx950 kernels.
//
// Host-pointer entry points stage through library-owned, grow-only device buffers on one private
// stream and block; *_dev entry points only enqueue on the caller's stream.  There is no CPU
// path in this library: without a HIP device every solve entry point fails with CRX_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <dlfcn.h>

#include <chrono>
#include <mutex>
#include <vector>

// RCCL: types only -- the entry points are resolved at run time (rccl_bind), libcrx does not link librccl.  A ROCm install without
// the RCCL development headers still builds the single-GPU library: the handful of types the five entry points need are declared
// here then (the stable NCCL 2 ABI: opaque communicator, 128-byte id, int-valued enums)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
}
#endif

#include "crx_kparams.h"

namespace {

thread_local char g_err[512] = "";
std::mutex g_mu;
int g_device = -1;
bool g_init = false;
hipStream_t g_stream = nullptr;

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(CRX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 2 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(CRX_ERR_HIP, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
DevBuf g_in, g_out;  // staging for the host-pointer entry points

// pinned host mirrors of g_in / g_out: a host-pointer call gathers ALL its inputs into one pinned block
// (one H2D copy), and scatters its outputs from one pinned block (one D2H copy) -- a batch-1 control step
// is launch-latency bound, and a dozen separate small copies cost more than the solve itself
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 2 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(CRX_ERR_HIP, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};
PinBuf g_hin, g_hout;
DevBuf g_trace;
int g_poison = 0;
int g_kkt_unscaled = 0;
int g_spec = 0;    // diagnostics (crx_debug_speculation): 0 = never (default), 1 = wherever the two-wave instantiation exists, 2 = that kernel with its second wave idle
int g_trace_rows = 0, g_trace_problem = 0;

int ensure_init() {
    if (g_init) return 0;
    return fail(CRX_ERR_NOT_INIT, "crx_init() has not been called (or failed)");
}

// staging plan of one host-pointer call: in()/out() carve the device buffers and their pinned mirrors
// at identical offsets; up() / down() move each block once
struct Stage {
    size_t oi = 0, oo = 0;
    struct Back { void* host; size_t off, bytes; };
    Back backs[16];
    int nb = 0;
    int reserve(size_t bytes_in, size_t bytes_out) {
        bytes_in += 20 * 256; bytes_out += 20 * 256;   // alignment slack for up to 20 arrays each
        if (int rc = g_in.ensure(bytes_in)) return rc;
        if (int rc = g_hin.ensure(bytes_in)) return rc;
        if (int rc = g_out.ensure(bytes_out)) return rc;
        return g_hout.ensure(bytes_out);
    }
    template <typename T>
    T* in(const T* src, size_t n) {   // src may be NULL (array not used by this call): zero-filled
        oi = (oi + 255) & ~size_t(255);
        T* dev = (T*)((char*)g_in.p + oi);
        if (n) {
            if (src) memcpy((char*)g_hin.p + oi, src, n * sizeof(T)); else memset((char*)g_hin.p + oi, 0, n * sizeof(T));
        }
        oi += n * sizeof(T);
        return dev;
    }
    template <typename T>
    T* out(T* host, size_t n) {       // host may be NULL (result not wanted)
        oo = (oo + 255) & ~size_t(255);
        T* dev = (T*)((char*)g_out.p + oo);
        backs[nb++] = Back{(void*)host, oo, n * sizeof(T)};
        oo += n * sizeof(T);
        return dev;
    }
    int up(hipStream_t st) {
        if (oi) HIP_TRY(hipMemcpyAsync(g_in.p, g_hin.p, oi, hipMemcpyHostToDevice, st));
        return 0;
    }
    int down(hipStream_t st) {
        if (oo) HIP_TRY(hipMemcpyAsync(g_hout.p, g_out.p, oo, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < nb; i++)
            if (backs[i].host && backs[i].bytes) memcpy(backs[i].host, (char*)g_hout.p + backs[i].off, backs[i].bytes);
        return 0;
    }
};

// bump allocator over a DevBuf
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* r = (T*)(base + off);
        off += n * sizeof(T);
        return r;
    }
};

int check_opts(const crx_ipm_opts& o) {
    if (!(o.tol > 0) || o.max_iter < 1 || !(o.mu_init > 0) || !(o.tau_min > 0 && o.tau_min < 1) ||
        !(o.slack_push > 0) || !(o.kappa_mu > 0 && o.kappa_mu < 1) || !(o.theta_mu > 1) || !(o.grad_scale_max > 0) ||
        o.slack_start < 0 || o.slack_start > 3 || !(o.dual_inf_tol > 0) || !(o.constr_viol_tol > 0) || !(o.compl_inf_tol > 0) || o.stall_iters < 1 || o.qp_method < 0 || o.qp_method > 1)
        return fail(CRX_ERR_ARG, "invalid crx_ipm_opts (a descriptor built for libcrx <= 0.3.x? crx_ipm_opts grew in 0.2 and in 0.4: include/crx.h)");
    return 0;
}

int fill_planner(crx_kparams& kp, const crx_planner_desc* d, int batch) {
    if (!d) return fail(CRX_ERR_ARG, "desc is NULL");
    if (d->N < 3 || d->N > CRX_MAX_N) return fail(CRX_ERR_ARG, "N=%d outside [3,%d]", d->N, CRX_MAX_N);
    if (batch < 0) return fail(CRX_ERR_ARG, "batch < 0");
    if (int rc = check_opts(d->opts)) return rc;
    memset(&kp, 0, sizeof(kp));
    kp.N = d->N; kp.batch = batch; kp.mode = 0; kp.n_obs_max = 0; kp.degree = 6;
    memcpy(kp.A, d->A, sizeof(kp.A)); memcpy(kp.B, d->B, sizeof(kp.B));
    kp.wq[4] = d->w_ref; kp.wq[5] = d->w_ref;
    kp.w_dey = d->w_dey; kp.w_prog = d->w_prog; kp.w_slack = 0.0;
    kp.delta_max = d->delta_max; kp.a_max = d->a_max; kp.v_min = -INFINITY; kp.v_max = d->vx_max; kp.ey_max = INFINITY;
    kp.alpha = 0.0; kp.margin = 0.0; kp.l_sum = 1.0; kp.w_sum = 1.0;
    kp.dt_ref = d->dt_ref; kp.fallback_gain = d->fallback_gain; kp.opts = d->opts;
    // reachability screen (crx_kernels.hip, set-up): reach_gain[j] = sum_{m < j} |e_ey' A^m B| (delta_max, a_max)'
    kp.reach_screen = d->opts.reach_screen ? 1 : 0;
    double w[6] = {0, 0, 0, 0, 0, 1}, acc = 0.0;
    kp.reach_gain[0] = 0.0;
    memcpy(kp.reach_row[0], w, sizeof(w));
    for (int j = 1; j < d->N; j++) {
        double v0 = 0.0, v1 = 0.0, wn[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; i++) { v0 += w[i] * d->B[i * 2]; v1 += w[i] * d->B[i * 2 + 1]; }
        acc += fabs(v0) * d->delta_max + fabs(v1) * d->a_max;
        kp.reach_gain[j] = acc;
        for (int a = 0; a < 6; a++)
            for (int i = 0; i < 6; i++) wn[a] += w[i] * d->A[i * 6 + a];
        memcpy(w, wn, sizeof(w));
        memcpy(kp.reach_row[j], w, sizeof(w));
    }
    return 0;
}

int fill_cbf(crx_kparams& kp, const crx_cbf_desc* d, int batch) {
    if (!d) return fail(CRX_ERR_ARG, "desc is NULL");
    if (d->N < 3 || d->N > CRX_MAX_N) return fail(CRX_ERR_ARG, "N=%d outside [3,%d]", d->N, CRX_MAX_N);
    if (d->n_obs_max < 0 || d->n_obs_max > CRX_MAX_OBS) return fail(CRX_ERR_ARG, "n_obs_max=%d outside [0,%d]", d->n_obs_max, CRX_MAX_OBS);
    if (d->degree < 2 || d->degree > 8 || (d->degree & 1)) return fail(CRX_ERR_ARG, "degree must be 2, 4, 6 or 8");
    if (batch < 0) return fail(CRX_ERR_ARG, "batch < 0");
    if (!(d->alpha > 0.0 && d->alpha <= 1.0)) return fail(CRX_ERR_ARG, "alpha outside (0,1]");
    if (int rc = check_opts(d->opts)) return rc;
    memset(&kp, 0, sizeof(kp));
    kp.N = d->N; kp.batch = batch; kp.mode = 1; kp.per_stage_target = d->per_stage_target ? 1 : 0;
    kp.n_obs_max = d->n_obs_max; kp.degree = d->degree;
    memcpy(kp.A, d->A, sizeof(kp.A)); memcpy(kp.B, d->B, sizeof(kp.B));
    memcpy(kp.wq, d->Q, sizeof(kp.wq)); kp.wr[0] = d->R[0]; kp.wr[1] = d->R[1];
    kp.w_dey = 0.0; kp.w_prog = 0.0; kp.w_slack = d->w_slack;
    kp.delta_max = d->delta_max; kp.a_max = d->a_max; kp.v_min = d->v_min; kp.v_max = d->v_max; kp.ey_max = d->ey_max;
    kp.alpha = d->alpha; kp.margin = d->margin; kp.l_sum = d->l_sum; kp.w_sum = d->w_sum;
    kp.dt_ref = 0.1; kp.fallback_gain = 1.1; kp.opts = d->opts;
    // reach of s and ey under the boxed inputs (crx_kernels.hip: slack start): sum_{m<j} |e' A^m B| (delta_max, a_max)'
    kp.slack_start = d->opts.slack_start;
    double ws[6] = {0, 0, 0, 0, 1, 0}, we[6] = {0, 0, 0, 0, 0, 1}, as = 0.0, ae = 0.0;
    kp.reach_s[0] = 0.0; kp.reach_gain[0] = 0.0;
    for (int j = 1; j <= d->N; j++) {
        double s0 = 0.0, s1 = 0.0, e0 = 0.0, e1 = 0.0, wn[6] = {0, 0, 0, 0, 0, 0}, en[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; i++) {
            s0 += ws[i] * d->B[i * 2]; s1 += ws[i] * d->B[i * 2 + 1];
            e0 += we[i] * d->B[i * 2]; e1 += we[i] * d->B[i * 2 + 1];
        }
        as += fabs(s0) * d->delta_max + fabs(s1) * d->a_max;
        ae += fabs(e0) * d->delta_max + fabs(e1) * d->a_max;
        kp.reach_s[j] = as; kp.reach_gain[j] = ae;
        for (int a = 0; a < 6; a++)
            for (int i = 0; i < 6; i++) { wn[a] += ws[i] * d->A[i * 6 + a]; en[a] += we[i] * d->A[i * 6 + a]; }
        memcpy(ws, wn, sizeof(ws)); memcpy(we, en, sizeof(we));
    }
    return 0;
}

int launch_solve(const crx_kparams& kp, int tmpl, hipStream_t st) {
    crx_kparams kq = kp;
    if (g_trace_rows != 0) { kq.trace = (double*)g_trace.p; kq.trace_problem = g_trace_problem; kq.trace_rows = g_trace_rows; }
    kq.poison = g_poison;
    kq.kkt_unscaled = g_kkt_unscaled;
    size_t lds = crx_solve_lds_bytes(kp.N, tmpl);
    if (lds > 160 * 1024) return fail(CRX_ERR_ARG, "N=%d with %d obstacles needs %zu B of LDS (> 160 KiB)", kp.N, tmpl, lds);
    // [r6] The TWO-WAVE instantiation (one obstacle slot, the reference's exponent, N = 12 / 10; crx_kernels.hip SPEC): a second wave per problem factorises
    // with the next entry of the inertia-correction schedule while the first tries the current one.  OPT-IN (crx_debug_speculation), not the default:
    // measured on the headline batch (profiles/r06_speculation.txt) a doomed attempt costs 1.7 us, not a sweep's 5.8 -- the recursion stops at the
    // first non-positive pivot -- so the longest healthy solve (34 iterations, 24 doomed attempts) gains 2.5 % alone and the 256-problem launch, which
    // pays two workgroup barriers per iteration in every problem, loses 1 % (0.4759 -> 0.4805 ms).
    const bool spec = tmpl == 1 && kq.mode == 1 && kq.degree == 6 && (kq.N == 12 || kq.N == 10) && g_spec > 0;
    kq.spec_idle = g_spec == 2;
    hipError_t e = spec ? crx_launch_solve_spec(kq, st) : crx_launch_solve(kq, tmpl, st);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "solver launch: %s", hipGetErrorString(e));
    return 0;
}

}  // namespace

extern "C" {

int crx_version(void) { return CRX_VERSION; }

const char* crx_last_error(void) { return g_err; }

int crx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int crx_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(CRX_ERR_NO_DEVICE, "no HIP device visible (%s); libcrx has no CPU back-end",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(CRX_ERR_ARG, "device %d outside [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    if (g_init && g_device == device) return CRX_OK;
    if (g_init && g_device >= 0) {
        // everything below belongs to the old device: release it there (the trace buffer included)
        (void)hipSetDevice(g_device);
        if (g_stream) { (void)hipStreamDestroy(g_stream); g_stream = nullptr; }
        g_in.release(); g_out.release(); g_hin.release(); g_hout.release(); g_trace.release();
        g_trace_rows = 0;
        HIP_TRY(hipSetDevice(device));
    }
    HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_device = device;
    g_init = true;
    return CRX_OK;
}

void crx_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return;
    (void)hipSetDevice(g_device);
    g_in.release(); g_out.release(); g_hin.release(); g_hout.release(); g_trace.release();
    g_trace_rows = 0;
    if (g_stream) (void)hipStreamDestroy(g_stream);
    g_stream = nullptr;
    g_init = false; g_device = -1;
}

// diagnostics (not in crx.h): record e_d, e_p, e_c, mu, alpha, alpha_dual, delta_w, accept-type per
// iteration of one problem of the following solves
int crx_trace_enable(int problem, int rows) {
    if (int rc = ensure_init()) return rc;
    g_trace_rows = 0;
    const bool sub = rows < 0;   // negative: report Riccati sub-phase cycles in slots 8..11
    if (rows < 0) rows = -rows;
    if (rows == 0) return CRX_OK;
    if (int rc = g_trace.ensure((size_t)rows * 16 * sizeof(double))) return rc;
    HIP_TRY(hipMemset(g_trace.p, 0, (size_t)rows * 16 * sizeof(double)));
    g_trace_rows = sub ? -rows : rows; g_trace_problem = problem;
    return CRX_OK;
}
int crx_trace_read(double* host, int rows) {
    { int cap = g_trace_rows < 0 ? -g_trace_rows : g_trace_rows; if (rows > cap) rows = cap; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, g_trace.p, (size_t)rows * 16 * sizeof(double), hipMemcpyDeviceToHost));
    return CRX_OK;
}

// diagnostics (not in crx.h): make the solver kernels fill their LDS slice with NaN before set-up, so that a read
// of LDS the kernel did not write shows up as a changed result (tests/test_gpu_parity.py::test_no_stale_lds_reads)
void crx_debug_poison_lds(int enable) { g_poison = enable != 0; }

// diagnostics (not in crx.h): the solver kernel reports, for every CONVERGED problem of the following planner / CBF solves, an UNSCALED KKT
// quantity of the returned iterate in kkt[] instead of the scaled error -- without IPOPT's s_d = max(100, ||nu||_1 / m) / 100 and with the
// gradient-based row scaling undone (rows in the reference's units).  mode 1: the max of the three; 2: the reduced Lagrangian gradient (IPOPT
// dual_inf); 3: the constraint violation (constr_viol); 4: the complementarity (compl_inf); 0: off.  Since 0.4.0 these are the quantities the
// termination test itself bounds (crx_ipm_opts.dual_inf_tol / constr_viol_tol / compl_inf_tol); bench.py reports each (kkt_unscaled_*_max).
// diagnostics (not in crx.h): the two-wave (speculating) instantiation of the one-obstacle solver kernel -- 0: never (default), 1: whenever it exists
// (A/B runs; tests/test_gpu_parity.py::test_speculating_wave_follows_the_sequential_schedule), 2: that kernel with its second wave left idle (separates
// "another kernel" from "another wave's result" when two runs disagree)
void crx_debug_speculation(int mode) { g_spec = mode < 0 ? 0 : (mode > 2 ? 1 : mode); }

void crx_debug_kkt_unscaled(int mode) { g_kkt_unscaled = (mode >= 0 && mode <= 4) ? mode : 0; }

// diagnostics (not in crx.h): the packed wave reductions and the DPP dot products of crx_wave.h on host data, in [4][64] -> out [16 + 3 * 64]
// (tests/test_gpu_parity.py::test_packed_wave_reductions)
int crx_debug_wave_reduce(const double* in, double* out) {
    if (int rc = ensure_init()) return rc;
    if (!in || !out) return fail(CRX_ERR_ARG, "NULL array argument");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    Stage sg;
    if (int rc = sg.reserve(256 * 8, 208 * 8)) return rc;
    double* din = sg.in(in, 256);
    double* dout = sg.out(out, 208);
    if (int rc = sg.up(g_stream)) return rc;
    hipError_t e = crx_launch_debug_reduce(din, dout, g_stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "debug reduce launch: %s", hipGetErrorString(e));
    return sg.down(g_stream);
}

// diagnostics (not in crx.h): LDS bytes one problem occupies (= one single-wave workgroup), for the
// "resident problems per CU" figure of bench.py.  kind 0: crx_solve_kernel (N, n_obs_max); 1: crx_lmpc_kernel (N, n_ss_max)
long crx_debug_lds_bytes(int kind, int N, int n) {
    return kind == 0 ? (long)crx_solve_lds_bytes(N, n) : (long)crx_lmpc_lds_bytes(N, n);
}

// diagnostics (not in crx.h): problems resident per CU as the runtime computes it (LDS AND registers); needs a GPU
int crx_debug_resident_per_cu(int kind, int N, int n) {
    if (ensure_init() != CRX_OK) return -1;
    return kind == 0 ? crx_solve_resident_per_cu(N, n) : crx_lmpc_resident_per_cu(N, n);
}

// ---- timers: a pair of HIP events per object, no global state (include/crx.h) -----------------------------------------
struct CrxTimer { hipEvent_t e0, e1; bool armed; };

int crx_timer_create(void** timer) {
    if (int rc = ensure_init()) return rc;
    if (!timer) return fail(CRX_ERR_ARG, "timer is NULL");
    HIP_TRY(hipSetDevice(g_device));
    CrxTimer* t = new CrxTimer{nullptr, nullptr, false};
    if (hipEventCreate(&t->e0) != hipSuccess || hipEventCreate(&t->e1) != hipSuccess) {
        if (t->e0) (void)hipEventDestroy(t->e0);
        delete t;
        return fail(CRX_ERR_HIP, "hipEventCreate failed");
    }
    *timer = t;
    return CRX_OK;
}
int crx_timer_destroy(void* timer) {
    CrxTimer* t = (CrxTimer*)timer;
    if (!t) return CRX_OK;
    (void)hipEventDestroy(t->e0); (void)hipEventDestroy(t->e1);
    delete t;
    return CRX_OK;
}
int crx_timer_begin(void* timer, void* stream) {
    CrxTimer* t = (CrxTimer*)timer;
    if (!t) return fail(CRX_ERR_ARG, "timer is NULL");
    t->armed = false;
    HIP_TRY(hipEventRecord(t->e0, (hipStream_t)stream));
    return CRX_OK;
}
int crx_timer_end(void* timer, void* stream) {
    CrxTimer* t = (CrxTimer*)timer;
    if (!t) return fail(CRX_ERR_ARG, "timer is NULL");
    HIP_TRY(hipEventRecord(t->e1, (hipStream_t)stream));
    t->armed = true;
    return CRX_OK;
}
double crx_timer_ms(void* timer) {
    CrxTimer* t = (CrxTimer*)timer;
    if (!t || !t->armed) return -1.0;
    float ms = 0.f;
    if (hipEventSynchronize(t->e1) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, t->e0, t->e1) != hipSuccess) return -1.0;
    return (double)ms;
}

void crx_ipm_opts_default(crx_ipm_opts* o) {
    o->tol = 1e-8; o->max_iter = 200; o->restore_iters = 50; o->mu_init = 0.1; o->kappa_eps = 10.0;
    o->kappa_mu = 0.2; o->theta_mu = 1.5; o->tau_min = 0.99; o->slack_push = 1e-2; o->grad_scale_max = 100.0;
    o->reach_screen = 1; o->slack_start = 2;
    o->dual_inf_tol = 1.0; o->constr_viol_tol = 1e-4; o->compl_inf_tol = 1e-4;   // IPOPT's defaults (the reference sets none: control.py:593)
    o->stall_iters = 100; o->qp_method = 0;
}

void crx_planner_desc_default(crx_planner_desc* d, int N, const double* A, const double* B) {
    memset(d, 0, sizeof(*d));
    d->N = N;
    memcpy(d->A, A, sizeof(d->A)); memcpy(d->B, B, sizeof(d->B));
    d->w_ref = 20.0; d->w_dey = 30.0; d->w_prog = 200.0; d->vx_max = 5.0; d->delta_max = 0.5; d->a_max = 1.5;
    d->dt_ref = 0.1; d->fallback_gain = 1.1;
    crx_ipm_opts_default(&d->opts);
}

void crx_cbf_desc_default(crx_cbf_desc* d, int N, int n_obs_max, const double* A, const double* B) {
    memset(d, 0, sizeof(*d));
    d->N = N; d->n_obs_max = n_obs_max; d->per_stage_target = 0; d->degree = 6;
    memcpy(d->A, A, sizeof(d->A)); memcpy(d->B, B, sizeof(d->B));
    const double Q[6] = {10.0, 0.0, 0.0, 4.0, 0.0, 40.0};
    memcpy(d->Q, Q, sizeof(Q)); d->R[0] = 0.1; d->R[1] = 0.1;
    d->delta_max = 0.5; d->a_max = 1.0; d->v_min = 0.0; d->v_max = 10.0; d->ey_max = 1.0;
    d->alpha = 0.8; d->margin = 0.2; d->l_sum = 0.4; d->w_sum = 0.2; d->w_slack = 1e4;
    crx_ipm_opts_default(&d->opts);
    // budgets by problem class (include/crx.h crx_ipm_opts.stall_iters): the short one-obstacle class is done after 30 iterations when
    // healthy, the three-obstacle N = 20 class needs 53..86
    if (N <= 12 && n_obs_max <= 1) { d->opts.stall_iters = 50; d->opts.restore_iters = 25; }
}

void crx_select_desc_default(crx_select_desc* d, int N, int n_veh_max, double lap_length) {
    d->N = N; d->n_veh_max = n_veh_max; d->veh_length = 0.4; d->veh_width = 0.2; d->lap_length = lap_length;
    d->w_prog = 10.0; d->w_coll = 100.0; d->w_switch = 100.0;
}

// ---- planner --------------------------------------------------------------------------------------
static int planner_solve_masked(const crx_planner_desc* d, int batch, const int32_t* active, int active_div, const double* x0,
                                const double* bez_s, const double* bez_ey, const double* ey_lb, const double* ey_ub, double* X,
                                double* U, double* cost, int32_t* status, double* kkt, int32_t* iters, void* stream);

int crx_planner_solve_dev(const crx_planner_desc* d, int batch, const double* x0, const double* bez_s,
                          const double* bez_ey, const double* ey_lb, const double* ey_ub, double* X, double* U,
                          double* cost, int32_t* status, double* kkt, int32_t* iters, void* stream) {
    return planner_solve_masked(d, batch, nullptr, 0, x0, bez_s, bez_ey, ey_lb, ey_ub, X, U, cost, status, kkt, iters, stream);
}

static int planner_solve_masked(const crx_planner_desc* d, int batch, const int32_t* active, int active_div, const double* x0,
                                const double* bez_s, const double* bez_ey, const double* ey_lb, const double* ey_ub, double* X,
                                double* U, double* cost, int32_t* status, double* kkt, int32_t* iters, void* stream) {
    if (int rc = ensure_init()) return rc;
    crx_kparams kp;
    if (int rc = fill_planner(kp, d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!x0 || !bez_s || !bez_ey || !ey_lb || !ey_ub || !X || !U || !cost || !status || !kkt || !iters)
        return fail(CRX_ERR_ARG, "NULL array argument");
    kp.x0 = x0; kp.bez_s = bez_s; kp.bez_ey = bez_ey; kp.ey_lb = ey_lb; kp.ey_ub = ey_ub;
    kp.X = X; kp.U = U; kp.sigma = nullptr; kp.cost = cost; kp.status = status; kp.kkt = kkt; kp.iters = iters;
    kp.active = active; kp.active_div = active_div;
    return launch_solve(kp, 0, (hipStream_t)stream);
}

int crx_planner_solve(const crx_planner_desc* d, int batch, const double* x0, const double* bez_s,
                      const double* bez_ey, const double* ey_lb, const double* ey_ub, double* X, double* U,
                      double* cost, int32_t* status, double* kkt, int32_t* iters) {
    if (int rc = ensure_init()) return rc;
    crx_kparams chk;
    if (int rc = fill_planner(chk, d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!x0 || !bez_s || !bez_ey || !ey_lb || !ey_ub || !X || !U || !cost || !status || !kkt || !iters)
        return fail(CRX_ERR_ARG, "NULL array argument");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    const size_t B = (size_t)batch, N = (size_t)d->N;
    const size_t n_x0 = B * 6, n_bz = B * (N + 1), n_lb = B * N, n_ub = B;
    const size_t n_X = B * (N + 1) * 6, n_U = B * N * 2;
    Stage sg;
    if (int rc = sg.reserve((n_x0 + 2 * n_bz + n_lb + n_ub) * 8, (n_X + n_U + 2 * B) * 8 + 2 * B * 4)) return rc;
    double* dx0 = sg.in(x0, n_x0); double* dbs = sg.in(bez_s, n_bz); double* dbe = sg.in(bez_ey, n_bz);
    double* dlb = sg.in(ey_lb, n_lb); double* dub = sg.in(ey_ub, n_ub);
    double* dX = sg.out(X, n_X); double* dU = sg.out(U, n_U); double* dc = sg.out(cost, B);
    double* dk = sg.out(kkt, B); int32_t* ds = sg.out(status, B); int32_t* di = sg.out(iters, B);
    if (int rc = sg.up(g_stream)) return rc;
    if (int rc = crx_planner_solve_dev(d, batch, dx0, dbs, dbe, dlb, dub, dX, dU, dc, ds, dk, di, g_stream)) return rc;
    return sg.down(g_stream);
}

// ---- MPC-CBF ---------------------------------------------------------------------------------------
int crx_cbf_solve_dev(const crx_cbf_desc* d, int batch, const double* x0, const double* xt, const double* obs_s,
                      const double* obs_ey, const double* lap_off, const int32_t* n_obs, double* X, double* U,
                      double* sigma, double* cost, int32_t* status, double* kkt, int32_t* iters, void* stream) {
    return crx_cbf_solve_masked_dev(d, batch, nullptr, x0, xt, obs_s, obs_ey, lap_off, n_obs, X, U, sigma, cost, status, kkt, iters, stream);
}

int crx_cbf_solve_masked_dev(const crx_cbf_desc* d, int batch, const int32_t* active, const double* x0, const double* xt,
                             const double* obs_s, const double* obs_ey, const double* lap_off, const int32_t* n_obs, double* X,
                             double* U, double* sigma, double* cost, int32_t* status, double* kkt, int32_t* iters, void* stream) {
    return crx_cbf_solve_ordered_dev(d, batch, active, nullptr, x0, xt, obs_s, obs_ey, lap_off, n_obs, nullptr, X, U, sigma, cost, status, kkt, iters, stream);
}

int crx_cbf_solve_dims_dev(const crx_cbf_desc* d, int batch, const int32_t* active, const double* x0, const double* xt,
                           const double* obs_s, const double* obs_ey, const double* lap_off, const int32_t* n_obs,
                           const double* obs_dims, double* X, double* U, double* sigma, double* cost, int32_t* status, double* kkt,
                           int32_t* iters, void* stream) {
    return crx_cbf_solve_ordered_dev(d, batch, active, nullptr, x0, xt, obs_s, obs_ey, lap_off, n_obs, obs_dims, X, U, sigma, cost, status, kkt, iters, stream);
}

int crx_cbf_solve_ordered_dev(const crx_cbf_desc* d, int batch, const int32_t* active, const int32_t* order, const double* x0,
                              const double* xt, const double* obs_s, const double* obs_ey, const double* lap_off,
                              const int32_t* n_obs, const double* obs_dims, double* X, double* U, double* sigma, double* cost,
                              int32_t* status, double* kkt, int32_t* iters, void* stream) {
    if (int rc = ensure_init()) return rc;
    crx_kparams kp;
    if (int rc = fill_cbf(kp, d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!x0 || !xt || !X || !U || !cost || !status || !kkt || !iters) return fail(CRX_ERR_ARG, "NULL array argument");
    if (d->n_obs_max > 0 && (!obs_s || !obs_ey || !lap_off || !n_obs || !sigma))
        return fail(CRX_ERR_ARG, "NULL obstacle array with n_obs_max > 0");
    kp.x0 = x0; kp.xt = xt; kp.obs_s = obs_s; kp.obs_ey = obs_ey; kp.lap_off = lap_off; kp.n_obs = n_obs;
    kp.X = X; kp.U = U; kp.sigma = sigma; kp.cost = cost; kp.status = status; kp.kkt = kkt; kp.iters = iters;
    kp.active = active; kp.order = order; kp.obs_dims = d->n_obs_max > 0 ? obs_dims : nullptr;
    return launch_solve(kp, d->n_obs_max, (hipStream_t)stream);
}

int crx_cbf_solve(const crx_cbf_desc* d, int batch, const double* x0, const double* xt, const double* obs_s,
                  const double* obs_ey, const double* lap_off, const int32_t* n_obs, double* X, double* U,
                  double* sigma, double* cost, int32_t* status, double* kkt, int32_t* iters) {
    return crx_cbf_solve_dims(d, batch, x0, xt, obs_s, obs_ey, lap_off, n_obs, nullptr, X, U, sigma, cost, status, kkt, iters);
}

int crx_cbf_solve_dims(const crx_cbf_desc* d, int batch, const double* x0, const double* xt, const double* obs_s,
                  const double* obs_ey, const double* lap_off, const int32_t* n_obs, const double* obs_dims, double* X, double* U,
                  double* sigma, double* cost, int32_t* status, double* kkt, int32_t* iters) {
    if (int rc = ensure_init()) return rc;
    crx_kparams chk;
    if (int rc = fill_cbf(chk, d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!x0 || !xt || !X || !U || !cost || !status || !kkt || !iters) return fail(CRX_ERR_ARG, "NULL array argument");
    const size_t B = (size_t)batch, N = (size_t)d->N, V = (size_t)d->n_obs_max;
    if (V > 0) {
        if (!obs_s || !obs_ey || !lap_off || !n_obs || !sigma) return fail(CRX_ERR_ARG, "NULL obstacle array with n_obs_max > 0");
        for (size_t b = 0; b < B; b++)
            if (n_obs[b] < 0 || n_obs[b] > (int)V) return fail(CRX_ERR_ARG, "n_obs[%zu]=%d outside [0,%zu]", b, n_obs[b], V);
        if (obs_dims)
            for (size_t b = 0; b < B; b++)
                for (int o = 0; o < n_obs[b]; o++)
                    if (!(obs_dims[(b * V + o) * 2] > 0.0) || !(obs_dims[(b * V + o) * 2 + 1] > 0.0))
                        return fail(CRX_ERR_ARG, "obs_dims[%zu][%d] must be positive", b, o);
    }
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    const size_t n_x0 = B * 6, n_xt = d->per_stage_target ? B * (N + 1) * 6 : B * 6, n_ob = B * V * (N + 1), n_lo = B * V;
    const size_t n_X = B * (N + 1) * 6, n_U = B * N * 2;
    Stage sg;
    if (int rc = sg.reserve((n_x0 + n_xt + 2 * n_ob + 3 * n_lo) * 8 + B * 4, (n_X + n_U + n_ob + 2 * B) * 8 + 2 * B * 4)) return rc;
    double* dx0 = sg.in(x0, n_x0); double* dxt = sg.in(xt, n_xt);
    double* dos = sg.in(obs_s, n_ob); double* doe = sg.in(obs_ey, n_ob); double* dlo = sg.in(lap_off, n_lo);
    double* ddm = (obs_dims && V > 0) ? sg.in(obs_dims, 2 * n_lo) : nullptr;
    int32_t* dno = sg.in(V > 0 ? n_obs : (const int32_t*)nullptr, B);
    double* dX = sg.out(X, n_X); double* dU = sg.out(U, n_U); double* dsg = sg.out(sigma, n_ob);
    double* dc = sg.out(cost, B); double* dk = sg.out(kkt, B); int32_t* ds = sg.out(status, B); int32_t* di = sg.out(iters, B);
    if (int rc = sg.up(g_stream)) return rc;
    if (int rc = crx_cbf_solve_dims_dev(d, batch, nullptr, dx0, dxt, dos, doe, dlo, dno, ddm, dX, dU, dsg, dc, ds, dk, di, g_stream)) return rc;
    return sg.down(g_stream);
}

// ---- selection -------------------------------------------------------------------------------------
static int check_select(const crx_select_desc* d, int n_scen) {
    if (!d) return fail(CRX_ERR_ARG, "desc is NULL");
    if (d->N < 1 || d->N > CRX_MAX_N || d->n_veh_max < 0 || d->n_veh_max > CRX_MAX_VEH || n_scen < 0)
        return fail(CRX_ERR_ARG, "bad selection dimensions");
    // the kernel wraps predicted s by lap_length (overtake_traj_planner.py:216-217): a zeroed descriptor must be a call
    // failure, not a device loop
    if (!(d->lap_length > 0.0) || !isfinite(d->lap_length)) return fail(CRX_ERR_ARG, "lap_length must be positive and finite");
    if (!(d->veh_length >= 0.0) || !(d->veh_width >= 0.0)) return fail(CRX_ERR_ARG, "vehicle dimensions must be non-negative");
    return 0;
}

int crx_select_dev(const crx_select_desc* d, int n_scen, const int32_t* n_veh, const double* X, const double* obs_s,
                   const double* obs_ey, const int32_t* old_flag, int32_t* flag, double* sel_cost, double* best_X,
                   void* stream) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_select(d, n_scen)) return rc;
    if (n_scen == 0) return CRX_OK;
    if (!n_veh || !X || !old_flag || !flag || !sel_cost || !best_X || (d->n_veh_max > 0 && (!obs_s || !obs_ey)))
        return fail(CRX_ERR_ARG, "NULL array argument");
    crx_select_kparams sp;
    sp.N = d->N; sp.V = d->n_veh_max; sp.n_scen = n_scen;
    sp.veh_length = d->veh_length; sp.veh_width = d->veh_width; sp.lap_length = d->lap_length;
    sp.w_prog = d->w_prog; sp.w_coll = d->w_coll; sp.w_switch = d->w_switch;
    sp.n_veh = n_veh; sp.X = X; sp.obs_s = obs_s; sp.obs_ey = obs_ey; sp.old_flag = old_flag;
    sp.flag = flag; sp.sel_cost = sel_cost; sp.best_X = best_X;
    hipError_t e = crx_launch_select(sp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "selection launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_select(const crx_select_desc* d, int n_scen, const int32_t* n_veh, const double* X, const double* obs_s,
               const double* obs_ey, const int32_t* old_flag, int32_t* flag, double* sel_cost, double* best_X) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_select(d, n_scen)) return rc;
    if (n_scen == 0) return CRX_OK;
    if (!n_veh || !X || !old_flag || !flag || !sel_cost || !best_X || (d->n_veh_max > 0 && (!obs_s || !obs_ey)))
        return fail(CRX_ERR_ARG, "NULL array argument");
    const size_t S = (size_t)n_scen, N = (size_t)d->N, V = (size_t)d->n_veh_max, R = V + 1;
    for (size_t s = 0; s < S; s++)
        if (n_veh[s] < 0 || n_veh[s] > (int)V) return fail(CRX_ERR_ARG, "n_veh[%zu]=%d outside [0,%zu]", s, n_veh[s], V);
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    const size_t n_X = S * R * (N + 1) * 6, n_ob = S * V * (N + 1), n_bX = S * (N + 1) * 6;
    Stage sg;
    if (int rc = sg.reserve((n_X + 2 * n_ob) * 8 + 2 * S * 4, (S * R + n_bX) * 8 + S * 4)) return rc;
    double* dX = sg.in(X, n_X); double* dos = sg.in(obs_s, n_ob); double* doe = sg.in(obs_ey, n_ob);
    int32_t* dnv = sg.in(n_veh, S); int32_t* dof = sg.in(old_flag, S);
    int32_t* dfl = sg.out(flag, S); double* dsc = sg.out(sel_cost, S * R); double* dbX = sg.out(best_X, n_bX);
    if (int rc = sg.up(g_stream)) return rc;
    if (int rc = crx_select_dev(d, n_scen, dnv, dX, dos, doe, dof, dfl, dsc, dbX, g_stream)) return rc;
    return sg.down(g_stream);
}

// ---- overtake path planner QP -----------------------------------------------------------------------
void crx_path_desc_default(crx_path_desc* d, int N, double alpha) {
    memset(d, 0, sizeof(*d));
    d->N = N; d->alpha = alpha; d->w_rate = 100.0;
    crx_ipm_opts_default(&d->opts);
}

static int fill_path(crx_path_kparams& pp, const crx_path_desc* d, int batch) {
    if (!d) return fail(CRX_ERR_ARG, "desc is NULL");
    if (d->N < 2 || d->N > CRX_MAX_N) return fail(CRX_ERR_ARG, "N=%d outside [2,%d]", d->N, CRX_MAX_N);
    if (!(d->alpha >= 0.0 && d->alpha <= 1.0) || !(d->w_rate >= 0.0)) return fail(CRX_ERR_ARG, "alpha outside [0,1] or w_rate < 0");
    if (batch < 0) return fail(CRX_ERR_ARG, "batch < 0");
    if (int rc = check_opts(d->opts)) return rc;
    memset(&pp, 0, sizeof(pp));
    pp.N = d->N; pp.batch = batch; pp.alpha = d->alpha; pp.w_rate = d->w_rate; pp.opts = d->opts;
    return 0;
}

int crx_path_solve_dev(const crx_path_desc* d, int batch, const double* opt, const double* bez, const double* lb,
                       const double* ub, const double* e0, const double* eN, double* E, double* cost, int32_t* status,
                       double* kkt, int32_t* iters, void* stream) {
    if (int rc = ensure_init()) return rc;
    crx_path_kparams pp;
    if (int rc = fill_path(pp, d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!opt || !bez || !lb || !ub || !e0 || !eN || !E || !cost || !status || !kkt || !iters) return fail(CRX_ERR_ARG, "NULL array argument");
    pp.opt = opt; pp.bez = bez; pp.lb = lb; pp.ub = ub; pp.e0 = e0; pp.eN = eN;
    pp.E = E; pp.cost = cost; pp.status = status; pp.kkt = kkt; pp.iters = iters;
    hipError_t e = crx_launch_path(pp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "path launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_path_solve(const crx_path_desc* d, int batch, const double* opt, const double* bez, const double* lb,
                   const double* ub, const double* e0, const double* eN, double* E, double* cost, int32_t* status,
                   double* kkt, int32_t* iters) {
    if (int rc = ensure_init()) return rc;
    crx_path_kparams chk;
    if (int rc = fill_path(chk, d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!opt || !bez || !lb || !ub || !e0 || !eN || !E || !cost || !status || !kkt || !iters) return fail(CRX_ERR_ARG, "NULL array argument");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    const size_t B = (size_t)batch, n1 = B * (size_t)(d->N + 1);
    Stage sg;
    if (int rc = sg.reserve((4 * n1 + 2 * B) * 8, (n1 + 2 * B) * 8 + 2 * B * 4)) return rc;
    double* dop = sg.in(opt, n1); double* dbz = sg.in(bez, n1); double* dlb = sg.in(lb, n1); double* dub = sg.in(ub, n1);
    double* d0 = sg.in(e0, B); double* dN = sg.in(eN, B);
    double* dE = sg.out(E, n1); double* dc = sg.out(cost, B); double* dk = sg.out(kkt, B);
    int32_t* ds = sg.out(status, B); int32_t* di = sg.out(iters, B);
    if (int rc = sg.up(g_stream)) return rc;
    if (int rc = crx_path_solve_dev(d, batch, dop, dbz, dlb, dub, d0, dN, dE, dc, ds, dk, di, g_stream)) return rc;
    return sg.down(g_stream);
}

// ---- plant ----------------------------------------------------------------------------------------
void crx_plant_desc_default(crx_plant_desc* d, int n_seg, double lap_length) {
    memset(d, 0, sizeof(*d));
    d->n_sub = 100; d->n_seg = n_seg; d->dt_sub = 0.001; d->lap_length = lap_length;
    d->m = 1.98; d->lf = 0.125; d->lr = 0.125; d->Iz = 0.024;
    d->Df = 0.8 * 1.98 * 9.81 / 2.0; d->Cf = 1.25; d->Bf = 1.0;
    d->Dr = 0.8 * 1.98 * 9.81 / 2.0; d->Cr = 1.25; d->Br = 1.0;
}

static int check_plant(const crx_plant_desc* d, int batch) {
    if (!d) return fail(CRX_ERR_ARG, "desc is NULL");
    if (d->n_sub < 0 || d->n_seg < 1 || d->n_seg > 64) return fail(CRX_ERR_ARG, "n_sub < 0 or n_seg outside [1,64]");
    if (!(d->lap_length > 0.0) || !(d->m > 0.0) || !(d->Iz > 0.0)) return fail(CRX_ERR_ARG, "lap_length, m, Iz must be positive");
    if (batch < 0) return fail(CRX_ERR_ARG, "batch < 0");
    return 0;
}

int crx_plant_step_dev(const crx_plant_desc* d, int batch, const double* track, const double* xglob, const double* xcurv,
                       const double* u, double* xglob_next, double* xcurv_next, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_plant(d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!track || !xglob || !xcurv || !u || !xglob_next || !xcurv_next) return fail(CRX_ERR_ARG, "NULL array argument");
    crx_plant_kparams pk;
    pk.d = *d; pk.batch = batch; pk.u_stride = 2; pk.wrap = 0; pk.track = track; pk.xglob = xglob; pk.xcurv = xcurv; pk.u = u;
    pk.xglob_next = xglob_next; pk.xcurv_next = xcurv_next; pk.laps = nullptr; pk.noise_z = nullptr;
    hipError_t e = crx_launch_plant(pk, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "plant launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_plant_step_wrap_dev(const crx_plant_desc* d, int batch, const double* track, const double* xglob, const double* xcurv,
                            const double* u, int u_stride, double* xglob_next, double* xcurv_next, int32_t* laps, void* stream) {
    return crx_plant_step_noise_dev(d, batch, track, xglob, xcurv, u, u_stride, nullptr, xglob_next, xcurv_next, laps, stream);
}

int crx_plant_step_noise_dev(const crx_plant_desc* d, int batch, const double* track, const double* xglob, const double* xcurv,
                             const double* u, int u_stride, const double* noise_z, double* xglob_next, double* xcurv_next,
                             int32_t* laps, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_plant(d, batch)) return rc;
    if (u_stride < 2) return fail(CRX_ERR_ARG, "u_stride < 2");
    if (batch == 0) return CRX_OK;
    if (!track || !xglob || !xcurv || !u || !xglob_next || !xcurv_next) return fail(CRX_ERR_ARG, "NULL array argument");
    crx_plant_kparams pk;
    pk.d = *d; pk.batch = batch; pk.u_stride = u_stride; pk.wrap = 1; pk.track = track; pk.xglob = xglob; pk.xcurv = xcurv; pk.u = u;
    pk.xglob_next = xglob_next; pk.xcurv_next = xcurv_next; pk.laps = laps; pk.noise_z = noise_z;
    hipError_t e = crx_launch_plant(pk, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "plant launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_cbf_prep_dev(int N, int V, double lap_length, double t, double dt, double safety_time, int batch,
                     const double* xcurv, const double* car_s0, const double* car_v, const double* car_ey,
                     double* obs_s, double* obs_ey, double* lap_off, int32_t* n_obs, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (N < 1 || N > CRX_MAX_N || V < 1 || V > CRX_MAX_OBS || batch < 0 || !(lap_length > 0.0)) return fail(CRX_ERR_ARG, "bad cbf prep dimensions");
    if (batch == 0) return CRX_OK;
    if (!xcurv || !car_s0 || !car_v || !car_ey || !obs_s || !obs_ey || !lap_off || !n_obs) return fail(CRX_ERR_ARG, "NULL array argument");
    crx_cbfprep_kparams cp;
    cp.N = N; cp.V = V; cp.batch = batch; cp.lap_length = lap_length; cp.t = t; cp.dt = dt; cp.safety_time = safety_time;
    cp.xcurv = xcurv; cp.car_s0 = car_s0; cp.car_v = car_v; cp.car_ey = car_ey;
    cp.obs_s = obs_s; cp.obs_ey = obs_ey; cp.lap_off = lap_off; cp.n_obs = n_obs;
    hipError_t e = crx_launch_cbfprep(cp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "cbf prep launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_plant_step(const crx_plant_desc* d, int batch, const double* track, const double* xglob, const double* xcurv,
                   const double* u, double* xglob_next, double* xcurv_next) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_plant(d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!track || !xglob || !xcurv || !u || !xglob_next || !xcurv_next) return fail(CRX_ERR_ARG, "NULL array argument");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    const size_t B = (size_t)batch, T = (size_t)d->n_seg * 6;
    Stage sg;
    if (int rc = sg.reserve((T + B * 14) * 8, B * 12 * 8)) return rc;
    double* dtr = sg.in(track, T); double* dg = sg.in(xglob, B * 6); double* dc = sg.in(xcurv, B * 6); double* du = sg.in(u, B * 2);
    double* dgn = sg.out(xglob_next, B * 6); double* dcn = sg.out(xcurv_next, B * 6);
    if (int rc = sg.up(g_stream)) return rc;
    if (int rc = crx_plant_step_dev(d, batch, dtr, dg, dc, du, dgn, dcn, g_stream)) return rc;
    return sg.down(g_stream);
}

// ---- planner host prep on the device ---------------------------------------------------------------
void crx_prep_desc_default(crx_prep_desc* d, int N, int n_veh_max, int n_opt, double track_width, double lap_length) {
    memset(d, 0, sizeof(*d));
    d->N = N; d->n_veh_max = n_veh_max; d->n_opt = n_opt;
    d->prediction_factor = 0.5; d->lookahead = 4.0; d->track_width = track_width; d->lap_length = lap_length;
    d->veh_length = 0.4; d->veh_width = 0.2; d->safety_margin = 0.15; d->dt_ref = 0.1;
}

static int fill_prep(crx_prep_kparams& pp, const crx_prep_desc* d, int n_scen) {
    if (!d) return fail(CRX_ERR_ARG, "desc is NULL");
    if (d->N < 3 || d->N > CRX_MAX_N) return fail(CRX_ERR_ARG, "N=%d outside [3,%d]", d->N, CRX_MAX_N);
    if (d->n_veh_max < 0 || d->n_veh_max > CRX_MAX_VEH) return fail(CRX_ERR_ARG, "n_veh_max=%d outside [0,%d]", d->n_veh_max, CRX_MAX_VEH);
    if (d->n_opt < 2) return fail(CRX_ERR_ARG, "n_opt < 2");
    if (n_scen < 0) return fail(CRX_ERR_ARG, "n_scen < 0");
    if (!(d->lap_length > 0.0) || !(d->track_width > 0.0)) return fail(CRX_ERR_ARG, "lap_length and track_width must be positive");
    memset(&pp, 0, sizeof(pp));
    pp.N = d->N; pp.V = d->n_veh_max; pp.n_scen = n_scen; pp.n_opt = d->n_opt;
    pp.prediction_factor = d->prediction_factor; pp.lookahead = d->lookahead; pp.track_width = d->track_width;
    pp.lap_length = d->lap_length; pp.veh_length = d->veh_length; pp.veh_width = d->veh_width;
    pp.safety_margin = d->safety_margin; pp.dt_ref = d->dt_ref;
    return 0;
}

int crx_planner_prep_dev(const crx_prep_desc* d, int n_scen, const double* x_wrapped, const double* x_raw,
                         const int32_t* n_veh, const double* veh_info, const double* max_dv, const double* obs_s,
                         const double* obs_ey, const double* opt_s, const double* opt_ey, double* x0, double* bez_s,
                         double* bez_ey, double* ey_lb, double* ey_ub, void* stream) {
    if (int rc = ensure_init()) return rc;
    crx_prep_kparams pp;
    if (int rc = fill_prep(pp, d, n_scen)) return rc;
    if (n_scen == 0) return CRX_OK;
    if (!x_wrapped || !x_raw || !n_veh || !max_dv || !opt_s || !opt_ey || !x0 || !bez_s || !bez_ey || !ey_lb || !ey_ub ||
        (d->n_veh_max > 0 && (!veh_info || !obs_s || !obs_ey)))
        return fail(CRX_ERR_ARG, "NULL array argument");
    pp.x_wrapped = x_wrapped; pp.x_raw = x_raw; pp.n_veh = n_veh; pp.veh_info = veh_info; pp.max_dv = max_dv;
    pp.obs_s = obs_s; pp.obs_ey = obs_ey; pp.opt_s = opt_s; pp.opt_ey = opt_ey;
    pp.x0 = x0; pp.bez_s = bez_s; pp.bez_ey = bez_ey; pp.ey_lb = ey_lb; pp.ey_ub = ey_ub;
    hipError_t e = crx_launch_prep(pp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "prep launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_planner_prep(const crx_prep_desc* d, int n_scen, const double* x_wrapped, const double* x_raw,
                     const int32_t* n_veh, const double* veh_info, const double* max_dv, const double* obs_s,
                     const double* obs_ey, const double* opt_s, const double* opt_ey, double* x0, double* bez_s,
                     double* bez_ey, double* ey_lb, double* ey_ub) {
    if (int rc = ensure_init()) return rc;
    crx_prep_kparams chk;
    if (int rc = fill_prep(chk, d, n_scen)) return rc;
    if (n_scen == 0) return CRX_OK;
    if (!x_wrapped || !x_raw || !n_veh || !max_dv || !opt_s || !opt_ey || !x0 || !bez_s || !bez_ey || !ey_lb || !ey_ub ||
        (d->n_veh_max > 0 && (!veh_info || !obs_s || !obs_ey)))
        return fail(CRX_ERR_ARG, "NULL array argument");
    for (int i = 0; i < n_scen; i++)
        if (n_veh[i] < 0 || n_veh[i] > d->n_veh_max) return fail(CRX_ERR_ARG, "n_veh[%d]=%d outside [0,%d]", i, n_veh[i], d->n_veh_max);
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    const size_t S = (size_t)n_scen, N = (size_t)d->N, V = (size_t)d->n_veh_max, R = V + 1, T = (size_t)d->n_opt;
    const size_t n_vi = S * V * 3, n_ob = S * V * (N + 1), n_bz = S * R * (N + 1), n_lb = S * R * N;
    Stage sg;
    if (int rc = sg.reserve((S * 13 + n_vi + 2 * n_ob + 2 * T) * 8 + S * 4, (S * R * 7 + 2 * n_bz + n_lb) * 8)) return rc;
    double* dxw = sg.in(x_wrapped, S * 6); double* dxr = sg.in(x_raw, S * 6); double* dmd = sg.in(max_dv, S);
    double* dvi = sg.in(veh_info, n_vi); double* dos = sg.in(obs_s, n_ob); double* doe = sg.in(obs_ey, n_ob);
    double* dts = sg.in(opt_s, T); double* dte = sg.in(opt_ey, T); int32_t* dnv = sg.in(n_veh, S);
    double* dx0 = sg.out(x0, S * R * 6); double* dbs = sg.out(bez_s, n_bz); double* dbe = sg.out(bez_ey, n_bz);
    double* dlb = sg.out(ey_lb, n_lb); double* dub = sg.out(ey_ub, S * R);
    if (int rc = sg.up(g_stream)) return rc;
    if (int rc = crx_planner_prep_dev(d, n_scen, dxw, dxr, dnv, dvi, dmd, dos, doe, dts, dte, dx0, dbs, dbe, dlb, dub, g_stream)) return rc;
    return sg.down(g_stream);
}

// ---- learning-MPC QP ------------------------------------------------------------------------------
void crx_lmpc_desc_default(crx_lmpc_desc* d, int N, int n_ss_max) {
    memset(d, 0, sizeof(*d));
    d->N = N; d->n_ss_max = n_ss_max;
    d->R[0] = 1.0; d->R[1] = 0.25; d->dR[0] = 4.0; d->dR[1] = 0.0; d->x_track[0] = 5.0;
    d->v_max = 10.0; d->ey_max = 1.0; d->delta_max = 0.5; d->a_max = 1.0; d->w_x0 = 1e4;
    crx_ipm_opts_default(&d->opts);
}

static int fill_lmpc(crx_lmpc_kparams& kp, const crx_lmpc_desc* d, int batch) {
    if (!d) return fail(CRX_ERR_ARG, "desc is NULL");
    if (d->N < 2 || d->N > CRX_LMPC_MAX_N) return fail(CRX_ERR_ARG, "N=%d outside [2,%d]", d->N, CRX_LMPC_MAX_N);
    if (d->n_ss_max < 1 || d->n_ss_max > CRX_MAX_SS) return fail(CRX_ERR_ARG, "n_ss_max=%d outside [1,%d]", d->n_ss_max, CRX_MAX_SS);
    if (batch < 0) return fail(CRX_ERR_ARG, "batch < 0");
    if (!(d->R[0] > 0.0 && d->R[1] > 0.0) || d->dR[0] < 0.0 || d->dR[1] < 0.0) return fail(CRX_ERR_ARG, "R must be positive, dR non-negative");
    for (int c = 0; c < 6; c++)
        if (d->Q[c] < 0.0) return fail(CRX_ERR_ARG, "Q must be non-negative");
    if (!(d->w_x0 > 0.0)) return fail(CRX_ERR_ARG, "w_x0 must be positive");
    if (int rc = check_opts(d->opts)) return rc;
    memset(&kp, 0, sizeof(kp));
    kp.N = d->N; kp.batch = batch; kp.n_ss_max = d->n_ss_max;
    memcpy(kp.Q, d->Q, sizeof(kp.Q)); memcpy(kp.R, d->R, sizeof(kp.R)); memcpy(kp.dR, d->dR, sizeof(kp.dR));
    memcpy(kp.x_track, d->x_track, sizeof(kp.x_track));
    kp.v_max = d->v_max; kp.ey_max = d->ey_max; kp.delta_max = d->delta_max; kp.a_max = d->a_max;
    kp.w_x0 = d->w_x0; kp.opts = d->opts;
    return 0;
}

// n_ss[] is validated on the device side of the boundary only by clamping is NOT acceptable (it indexes
// LDS), so the host entry point checks it; the _dev variant documents the precondition 1 <= n_ss <= n_ss_max.
int crx_lmpc_solve_dev(const crx_lmpc_desc* d, int batch, const double* x0, const double* u_old, const double* A,
                       const double* B, const double* C, const double* ss, const double* qfun, const int32_t* n_ss,
                       double* X, double* U, double* lambda, double* cost, int32_t* status, double* kkt,
                       int32_t* iters, void* stream) {
    return crx_lmpc_solve_masked_dev(d, batch, nullptr, x0, u_old, A, B, C, ss, qfun, n_ss, X, U, lambda, cost, status, kkt, iters, stream);
}

int crx_lmpc_solve_masked_dev(const crx_lmpc_desc* d, int batch, const int32_t* active, const double* x0, const double* u_old,
                              const double* A, const double* B, const double* C, const double* ss, const double* qfun,
                              const int32_t* n_ss, double* X, double* U, double* lambda, double* cost, int32_t* status,
                              double* kkt, int32_t* iters, void* stream) {
    return crx_lmpc_solve_ordered_dev(d, batch, active, nullptr, x0, u_old, A, B, C, ss, qfun, n_ss, X, U, lambda, cost, status, kkt, iters, stream);
}

int crx_lmpc_solve_ordered_dev(const crx_lmpc_desc* d, int batch, const int32_t* active, const int32_t* order, const double* x0,
                               const double* u_old, const double* A, const double* B, const double* C, const double* ss,
                               const double* qfun, const int32_t* n_ss, double* X, double* U, double* lambda, double* cost,
                               int32_t* status, double* kkt, int32_t* iters, void* stream) {
    if (int rc = ensure_init()) return rc;
    crx_lmpc_kparams kp;
    if (int rc = fill_lmpc(kp, d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!x0 || !u_old || !A || !B || !C || !ss || !qfun || !n_ss || !X || !U || !lambda || !cost || !status || !kkt || !iters)
        return fail(CRX_ERR_ARG, "NULL array argument");
    kp.x0 = x0; kp.u_old = u_old; kp.A = A; kp.B = B; kp.C = C; kp.ss = ss; kp.qfun = qfun; kp.n_ss = n_ss;
    kp.X = X; kp.U = U; kp.lambda = lambda; kp.cost = cost; kp.status = status; kp.kkt = kkt; kp.iters = iters;
    kp.active = active; kp.order = order; kp.reach_screen = d->opts.reach_screen ? 1 : 0;
    if (g_trace_rows > 0) { kp.trace = (double*)g_trace.p; kp.trace_problem = g_trace_problem; kp.trace_rows = g_trace_rows; }
    kp.poison = g_poison;
    hipError_t e = crx_launch_lmpc(kp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "lmpc launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_lmpc_solve(const crx_lmpc_desc* d, int batch, const double* x0, const double* u_old, const double* A,
                   const double* B, const double* C, const double* ss, const double* qfun, const int32_t* n_ss,
                   double* X, double* U, double* lambda, double* cost, int32_t* status, double* kkt, int32_t* iters) {
    if (int rc = ensure_init()) return rc;
    crx_lmpc_kparams chk;
    if (int rc = fill_lmpc(chk, d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!x0 || !u_old || !A || !B || !C || !ss || !qfun || !n_ss || !X || !U || !lambda || !cost || !status || !kkt || !iters)
        return fail(CRX_ERR_ARG, "NULL array argument");
    for (int b = 0; b < batch; b++)
        if (n_ss[b] < 1 || n_ss[b] > d->n_ss_max) return fail(CRX_ERR_ARG, "n_ss[%d]=%d outside [1,%d]", b, n_ss[b], d->n_ss_max);
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    const size_t Bn = (size_t)batch, N = (size_t)d->N, M = (size_t)d->n_ss_max;
    const size_t n_A = Bn * N * 36, n_B = Bn * N * 12, n_C = Bn * N * 6, n_ss_ = Bn * 6 * M, n_q = Bn * M;
    const size_t n_X = Bn * (N + 1) * 6, n_U = Bn * N * 2;
    Stage sg;
    if (int rc = sg.reserve((Bn * 8 + n_A + n_B + n_C + n_ss_ + n_q) * 8 + Bn * 4, (n_X + n_U + n_q + 2 * Bn) * 8 + 2 * Bn * 4)) return rc;
    double* dx0 = sg.in(x0, Bn * 6); double* duo = sg.in(u_old, Bn * 2);
    double* dA = sg.in(A, n_A); double* dB = sg.in(B, n_B); double* dC = sg.in(C, n_C);
    double* dss = sg.in(ss, n_ss_); double* dq = sg.in(qfun, n_q); int32_t* dn = sg.in(n_ss, Bn);
    double* dX = sg.out(X, n_X); double* dU = sg.out(U, n_U); double* dl = sg.out(lambda, n_q);
    double* dc = sg.out(cost, Bn); double* dk = sg.out(kkt, Bn); int32_t* ds = sg.out(status, Bn); int32_t* di = sg.out(iters, Bn);
    if (int rc = sg.up(g_stream)) return rc;
    if (int rc = crx_lmpc_solve_dev(d, batch, dx0, duo, dA, dB, dC, dss, dq, dn, dX, dU, dl, dc, ds, dk, di, g_stream)) return rc;
    return sg.down(g_stream);
}

// ---- planner front (interest test, partial sort, vehicle infos) on the device ------------------------
void crx_scene_desc_default(crx_scene_desc* d, int N, int n_all_max, int n_veh_max, double lap_length) {
    memset(d, 0, sizeof(*d));
    d->N = N; d->n_all_max = n_all_max; d->n_veh_max = n_veh_max;
    d->safety_factor = 4.5; d->prediction_factor = 0.5; d->veh_length = 0.4; d->lap_length = lap_length;
}

static int check_scene(const crx_scene_desc* d, int n_scen) {
    if (!d) return fail(CRX_ERR_ARG, "desc is NULL");
    if (d->N < 1 || d->N > CRX_MAX_N) return fail(CRX_ERR_ARG, "N=%d outside [1,%d]", d->N, CRX_MAX_N);
    if (d->n_all_max < 1 || d->n_all_max > 64) return fail(CRX_ERR_ARG, "n_all_max=%d outside [1,64]", d->n_all_max);
    if (d->n_veh_max < 1 || d->n_veh_max > CRX_MAX_VEH) return fail(CRX_ERR_ARG, "n_veh_max=%d outside [1,%d]", d->n_veh_max, CRX_MAX_VEH);
    if (!(d->lap_length > 0.0) || !isfinite(d->lap_length)) return fail(CRX_ERR_ARG, "lap_length must be positive and finite");
    if (n_scen < 0) return fail(CRX_ERR_ARG, "n_scen < 0");
    return 0;
}

int crx_planner_scene_dev(const crx_scene_desc* d, int n_scen, const double* ego_xcurv, const int32_t* n_all,
                          const double* veh_xcurv, const double* pred_s, const double* pred_ey, int32_t* n_veh,
                          int32_t* overflow, int32_t* order, double* veh_info, double* max_dv, double* obs_s, double* obs_ey,
                          void* stream) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_scene(d, n_scen)) return rc;
    if (n_scen == 0) return CRX_OK;
    if (!ego_xcurv || !n_all || !veh_xcurv || !pred_s || !pred_ey || !n_veh || !overflow || !order || !veh_info || !max_dv || !obs_s || !obs_ey)
        return fail(CRX_ERR_ARG, "NULL array argument");
    crx_scene_kparams sp;
    sp.d = *d; sp.n_scen = n_scen; sp.ego_xcurv = ego_xcurv; sp.n_all = n_all; sp.veh_xcurv = veh_xcurv; sp.pred_s = pred_s;
    sp.pred_ey = pred_ey; sp.n_veh = n_veh; sp.overflow = overflow; sp.order = order; sp.veh_info = veh_info; sp.max_dv = max_dv;
    sp.obs_s = obs_s; sp.obs_ey = obs_ey;
    hipError_t e = crx_launch_scene(sp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "scene launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_planner_scene(const crx_scene_desc* d, int n_scen, const double* ego_xcurv, const int32_t* n_all, const double* veh_xcurv,
                      const double* pred_s, const double* pred_ey, int32_t* n_veh, int32_t* overflow, int32_t* order,
                      double* veh_info, double* max_dv, double* obs_s, double* obs_ey) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_scene(d, n_scen)) return rc;
    if (n_scen == 0) return CRX_OK;
    if (!ego_xcurv || !n_all || !veh_xcurv || !pred_s || !pred_ey || !n_veh || !overflow || !order || !veh_info || !max_dv || !obs_s || !obs_ey)
        return fail(CRX_ERR_ARG, "NULL array argument");
    for (int i = 0; i < n_scen; i++)
        if (n_all[i] < 0 || n_all[i] > d->n_all_max) return fail(CRX_ERR_ARG, "n_all[%d]=%d outside [0,%d]", i, n_all[i], d->n_all_max);
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    const size_t S = (size_t)n_scen, VA = (size_t)d->n_all_max, V = (size_t)d->n_veh_max, N1 = (size_t)d->N + 1;
    Stage sg;
    if (int rc = sg.reserve((S * 6 + S * VA * 6 + 2 * S * VA * N1) * 8 + S * 4, (S * V * 3 + S + 2 * S * V * N1) * 8 + (2 * S + S * V) * 4)) return rc;
    double* dego = sg.in(ego_xcurv, S * 6); double* dvx = sg.in(veh_xcurv, S * VA * 6);
    double* dps = sg.in(pred_s, S * VA * N1); double* dpe = sg.in(pred_ey, S * VA * N1); int32_t* dna = sg.in(n_all, S);
    int32_t* dnv = sg.out(n_veh, S); int32_t* dov = sg.out(overflow, S); int32_t* dor = sg.out(order, S * V);
    double* dvi = sg.out(veh_info, S * V * 3); double* dmd = sg.out(max_dv, S);
    double* dos = sg.out(obs_s, S * V * N1); double* doe = sg.out(obs_ey, S * V * N1);
    if (int rc = sg.up(g_stream)) return rc;
    if (int rc = crx_planner_scene_dev(d, n_scen, dego, dna, dvx, dps, dpe, dnv, dov, dor, dvi, dmd, dos, doe, g_stream)) return rc;
    return sg.down(g_stream);
}

// ---- inputs of the tracking NLP from the planner's outputs, on the device ----------------------------
int crx_track_prep_dev(int N, int V, double lap_length, double safety_time, double dt_ref, int batch, const double* x,
                       const int32_t* n_veh, const double* obs_s_in, const double* obs_ey_in, const double* traj, double* xt,
                       double* obs_s, double* obs_ey, double* lap_off, int32_t* n_obs, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (N < 1 || N > CRX_MAX_N || V < 1 || V > CRX_MAX_OBS || batch < 0 || !(lap_length > 0.0) || !isfinite(lap_length))
        return fail(CRX_ERR_ARG, "bad track prep dimensions");
    if (batch == 0) return CRX_OK;
    if (!x || !n_veh || !obs_s_in || !obs_ey_in || !traj || !xt || !obs_s || !obs_ey || !lap_off || !n_obs) return fail(CRX_ERR_ARG, "NULL array argument");
    crx_trackprep_kparams tp;
    tp.N = N; tp.V = V; tp.batch = batch; tp.lap_length = lap_length; tp.safety_time = safety_time; tp.dt_ref = dt_ref;
    tp.x = x; tp.n_veh = n_veh; tp.obs_s_in = obs_s_in; tp.obs_ey_in = obs_ey_in; tp.traj = traj;
    tp.xt = xt; tp.obs_s = obs_s; tp.obs_ey = obs_ey; tp.lap_off = lap_off; tp.n_obs = n_obs;
    hipError_t e = crx_launch_trackprep(tp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "track prep launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

// ---- learning-MPC host prep on the device ------------------------------------------------------------
void crx_lmpcprep_desc_default(crx_lmpcprep_desc* d, int N, int n_points, int n_laps, int n_seg, double dt, double lap_length) {
    memset(d, 0, sizeof(*d));
    d->N = N; d->n_points = n_points; d->n_laps = n_laps; d->n_ss_per_lap = 22; d->n_ss_laps = 2; d->max_neighbours = 40;
    d->n_seg = n_seg; d->shift = 0; d->bandwidth = 5.0;
    d->scale[0] = 0.1; d->scale[1] = d->scale[2] = d->scale[3] = d->scale[4] = 1.0;
    d->dt = dt; d->lap_length = lap_length;
}

static int check_lmpcprep(const crx_lmpcprep_desc* d, int batch) {
    if (!d) return fail(CRX_ERR_ARG, "desc is NULL");
    if (d->N < 2 || d->N > CRX_LMPC_MAX_N) return fail(CRX_ERR_ARG, "N=%d outside [2,%d]", d->N, CRX_LMPC_MAX_N);
    if (d->n_points < 2 || d->n_points > 65535 || d->n_laps < 2) return fail(CRX_ERR_ARG, "n_points outside [2,65535] or n_laps < 2");
    if (d->n_ss_laps < 1 || d->n_ss_laps > 2 || d->n_ss_per_lap < 1 || d->n_ss_laps * d->n_ss_per_lap > CRX_MAX_SS)
        return fail(CRX_ERR_ARG, "n_ss_laps outside [1,2] or more than %d safe-set points", CRX_MAX_SS);
    if (d->max_neighbours < 1 || d->max_neighbours > 64) return fail(CRX_ERR_ARG, "max_neighbours outside [1,64]");
    if (d->n_seg < 1 || !(d->bandwidth > 0.0) || !(d->dt > 0.0) || !(d->lap_length > 0.0) || !isfinite(d->lap_length))
        return fail(CRX_ERR_ARG, "n_seg, bandwidth, dt and lap_length must be positive");
    if (batch < 0) return fail(CRX_ERR_ARG, "batch < 0");
    if (crx_lmpcprep_lds_bytes(d->n_points) > 160 * 1024) return fail(CRX_ERR_ARG, "n_points=%d needs more than 160 KiB of LDS", d->n_points);
    return 0;
}

int crx_lmpc_prep_dev(const crx_lmpcprep_desc* d, int batch, const double* ss_xcurv, const double* u_ss, const double* qfun,
                      const int32_t* time_ss, const int32_t* iter, const double* x, const double* lin_points,
                      const double* lin_input, int from_plan, const double* track, double* A, double* B, double* C,
                      double* ss_sel, double* q_sel, int32_t* status, void* stream) {
    return crx_lmpc_prep_masked_dev(d, batch, nullptr, ss_xcurv, u_ss, qfun, time_ss, iter, x, lin_points, lin_input, from_plan, track, A, B, C,
                                    ss_sel, q_sel, status, stream);
}

int crx_lmpc_prep_masked_dev(const crx_lmpcprep_desc* d, int batch, const int32_t* active, const double* ss_xcurv, const double* u_ss,
                             const double* qfun, const int32_t* time_ss, const int32_t* iter, const double* x, const double* lin_points,
                             const double* lin_input, int from_plan, const double* track, double* A, double* B, double* C,
                             double* ss_sel, double* q_sel, int32_t* status, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_lmpcprep(d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!ss_xcurv || !u_ss || !qfun || !time_ss || !iter || !x || !lin_points || !lin_input || !track || !A || !B || !C || !ss_sel ||
        !q_sel || !status)
        return fail(CRX_ERR_ARG, "NULL array argument");
    crx_lmpcprep_kparams kp;
    kp.d = *d; kp.batch = batch; kp.from_plan = from_plan ? 1 : 0;
    kp.ss_xcurv = ss_xcurv; kp.u_ss = u_ss; kp.qfun = qfun; kp.time_ss = time_ss; kp.iter = iter; kp.x = x;
    kp.lin_points = lin_points; kp.lin_input = lin_input; kp.track = track;
    kp.A = A; kp.B = B; kp.C = C; kp.ss_sel = ss_sel; kp.q_sel = q_sel; kp.status = status; kp.active = active;
    hipError_t e = crx_launch_lmpcprep(kp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "lmpc prep launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_lmpc_prep(const crx_lmpcprep_desc* d, int batch, const double* ss_xcurv, const double* u_ss, const double* qfun,
                  const int32_t* time_ss, const int32_t* iter, const double* x, const double* lin_points,
                  const double* lin_input, int from_plan, const double* track, double* A, double* B, double* C,
                  double* ss_sel, double* q_sel, int32_t* status) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_lmpcprep(d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!ss_xcurv || !u_ss || !qfun || !time_ss || !iter || !x || !lin_points || !lin_input || !track || !A || !B || !C || !ss_sel ||
        !q_sel || !status)
        return fail(CRX_ERR_ARG, "NULL array argument");
    const size_t Bn = (size_t)batch, N = (size_t)d->N, P = (size_t)d->n_points, L = (size_t)d->n_laps;
    const size_t M = (size_t)d->n_ss_per_lap * d->n_ss_laps;
    for (size_t b = 0; b < Bn; b++) {
        if (iter[b] < 2 || iter[b] > (int)L) return fail(CRX_ERR_ARG, "iter[%zu]=%d outside [2,%zu]", b, iter[b], L);
        for (int k = 0; k < 2; k++) {
            const int t = time_ss[b * L + iter[b] - 2 + k];
            if (t < 2 || t > (int)P) return fail(CRX_ERR_ARG, "time_ss of a lap used by race %zu is %d, outside [2,%zu]", b, t, P);
        }
    }
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    Stage sg;
    if (int rc = sg.reserve((Bn * L * P * 9 + Bn * (6 + (N + 1) * 6 + N * 2) + (size_t)d->n_seg * 6) * 8 + Bn * (L + 1) * 4,
                            (Bn * N * 54 + Bn * 7 * M) * 8 + Bn * 4)) return rc;
    double* dss = sg.in(ss_xcurv, Bn * L * P * 6); double* dus = sg.in(u_ss, Bn * L * P * 2); double* dqf = sg.in(qfun, Bn * L * P);
    double* dx = sg.in(x, Bn * 6); double* dlp = sg.in(lin_points, Bn * (N + 1) * 6); double* dli = sg.in(lin_input, Bn * N * 2);
    double* dtr = sg.in(track, (size_t)d->n_seg * 6);
    int32_t* dts = sg.in(time_ss, Bn * L); int32_t* dit = sg.in(iter, Bn);
    double* dA = sg.out(A, Bn * N * 36); double* dB = sg.out(B, Bn * N * 12); double* dC = sg.out(C, Bn * N * 6);
    double* dsel = sg.out(ss_sel, Bn * 6 * M); double* dq = sg.out(q_sel, Bn * M); int32_t* dst = sg.out(status, Bn);
    if (int rc = sg.up(g_stream)) return rc;
    // A, B, C are in/out: a singular stage leaves its three regression rows untouched (include/crx.h), so the staging copies
    // start from the caller's arrays, not from whatever an earlier call left in the staging buffer
    HIP_TRY(hipMemcpyAsync(dA, A, Bn * N * 36 * sizeof(double), hipMemcpyHostToDevice, g_stream));
    HIP_TRY(hipMemcpyAsync(dB, B, Bn * N * 12 * sizeof(double), hipMemcpyHostToDevice, g_stream));
    HIP_TRY(hipMemcpyAsync(dC, C, Bn * N * 6 * sizeof(double), hipMemcpyHostToDevice, g_stream));
    if (int rc = crx_lmpc_prep_dev(d, batch, dss, dus, dqf, dts, dit, dx, dlp, dli, from_plan, dtr, dA, dB, dC, dsel, dq, dst, g_stream)) return rc;
    return sg.down(g_stream);
}

int crx_lmpc_addpoint_dev(const crx_lmpcprep_desc* d, int batch, double* ss_xcurv, double* u_ss, const int32_t* time_ss,
                          const int32_t* iter, const int32_t* step, const double* x, const double* u, int u_stride, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_lmpcprep(d, batch)) return rc;
    if (u_stride < 2) return fail(CRX_ERR_ARG, "u_stride < 2");
    if (batch == 0) return CRX_OK;
    if (!ss_xcurv || !u_ss || !time_ss || !iter || !step || !x || !u) return fail(CRX_ERR_ARG, "NULL array argument");
    hipError_t e = crx_launch_lmpc_addpoint(*d, batch, ss_xcurv, u_ss, time_ss, iter, step, x, u, u_stride, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "lmpc addpoint launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_lmpc_addtraj_dev(const crx_lmpcprep_desc* d, int batch, const int32_t* crossed, double* log_x, const double* log_u,
                         int32_t* n_log, double* ss_xcurv, double* u_ss, double* qfun, int32_t* time_ss, int32_t* iter,
                         int32_t* step, const double* x, int32_t* status, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (int rc = check_lmpcprep(d, batch)) return rc;
    if (batch == 0) return CRX_OK;
    if (!crossed || !log_x || !log_u || !n_log || !ss_xcurv || !u_ss || !qfun || !time_ss || !iter || !step || !x || !status)
        return fail(CRX_ERR_ARG, "NULL array argument");
    hipError_t e = crx_launch_lmpc_addtraj(*d, batch, crossed, log_x, log_u, n_log, ss_xcurv, u_ss, qfun, time_ss, iter, step, x, status,
                                           (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "lmpc addtraj launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

// ---- fused planner step ------------------------------------------------------------------------------
int crx_planner_plan_dev(const crx_planner_desc* d, const crx_select_desc* sd, int n_scen, const double* x0,
                         const double* bez_s, const double* bez_ey, const double* ey_lb, const double* ey_ub,
                         const int32_t* n_veh, const double* obs_s, const double* obs_ey, const int32_t* old_flag,
                         double* X, double* U, double* cost, int32_t* status, double* kkt, int32_t* iters,
                         int32_t* flag, double* sel_cost, double* best_X, void* stream) {
    return crx_planner_plan_masked_dev(d, sd, n_scen, nullptr, x0, bez_s, bez_ey, ey_lb, ey_ub, n_veh, obs_s, obs_ey, old_flag, X, U, cost,
                                       status, kkt, iters, flag, sel_cost, best_X, stream);
}

int crx_planner_plan_masked_dev(const crx_planner_desc* d, const crx_select_desc* sd, int n_scen, const int32_t* active,
                                const double* x0, const double* bez_s, const double* bez_ey, const double* ey_lb,
                                const double* ey_ub, const int32_t* n_veh, const double* obs_s, const double* obs_ey,
                                const int32_t* old_flag, double* X, double* U, double* cost, int32_t* status, double* kkt,
                                int32_t* iters, int32_t* flag, double* sel_cost, double* best_X, void* stream) {
    if (!d || !sd) return fail(CRX_ERR_ARG, "desc is NULL");
    if (sd->N != d->N) return fail(CRX_ERR_ARG, "planner and selection horizons differ");
    if (int rc = check_select(sd, n_scen)) return rc;   // before R = n_veh_max + 1 sizes the QP launch
    const int R = sd->n_veh_max + 1;
    if ((long long)n_scen * R > 0x7fffffffLL) return fail(CRX_ERR_ARG, "n_scen * (n_veh_max + 1) overflows int");
    if (int rc = planner_solve_masked(d, n_scen * R, active, R, x0, bez_s, bez_ey, ey_lb, ey_ub, X, U, cost, status, kkt, iters, stream)) return rc;
    return crx_select_dev(sd, n_scen, n_veh, X, obs_s, obs_ey, old_flag, flag, sel_cost, best_X, stream);
}

int crx_planner_plan(const crx_planner_desc* d, const crx_select_desc* sd, int n_scen, const double* x0,
                     const double* bez_s, const double* bez_ey, const double* ey_lb, const double* ey_ub,
                     const int32_t* n_veh, const double* obs_s, const double* obs_ey, const int32_t* old_flag,
                     double* X, double* U, double* cost, int32_t* status, double* kkt, int32_t* iters,
                     int32_t* flag, double* sel_cost, double* best_X) {
    if (!d || !sd) return fail(CRX_ERR_ARG, "desc is NULL");
    if (sd->N != d->N) return fail(CRX_ERR_ARG, "planner and selection horizons differ");
    if (n_scen < 0) return fail(CRX_ERR_ARG, "n_scen < 0");
    if (int rc = ensure_init()) return rc;
    crx_kparams chk;
    if (int rc = fill_planner(chk, d, 0)) return rc;
    if (int rc = check_select(sd, n_scen)) return rc;
    if (n_scen == 0) return CRX_OK;
    if (!x0 || !bez_s || !bez_ey || !ey_lb || !ey_ub || !n_veh || !old_flag || !X || !U || !cost || !status || !kkt || !iters ||
        !flag || !sel_cost || !best_X || (sd->n_veh_max > 0 && (!obs_s || !obs_ey)))
        return fail(CRX_ERR_ARG, "NULL array argument");
    const size_t S = (size_t)n_scen, N = (size_t)d->N, V = (size_t)sd->n_veh_max, R = V + 1, B = S * R;
    for (size_t i = 0; i < S; i++)
        if (n_veh[i] < 0 || n_veh[i] > (int)V) return fail(CRX_ERR_ARG, "n_veh[%zu]=%d outside [0,%zu]", i, n_veh[i], V);
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g_device));
    // one staging round trip for the whole step; X stays on the device between the two kernels
    const size_t n_x0 = B * 6, n_bz = B * (N + 1), n_lb = B * N, n_ob = S * V * (N + 1);
    const size_t n_X = B * (N + 1) * 6, n_U = B * N * 2, n_bX = S * (N + 1) * 6;
    Stage sg;
    if (int rc = sg.reserve((n_x0 + 2 * n_bz + n_lb + B + 2 * n_ob) * 8 + 2 * S * 4,
                            (n_X + n_U + 2 * B + S * R + n_bX) * 8 + 2 * B * 4 + S * 4)) return rc;
    double* dx0 = sg.in(x0, n_x0); double* dbs = sg.in(bez_s, n_bz); double* dbe = sg.in(bez_ey, n_bz);
    double* dlb = sg.in(ey_lb, n_lb); double* dub = sg.in(ey_ub, B);
    double* dos = sg.in(obs_s, n_ob); double* doe = sg.in(obs_ey, n_ob);
    int32_t* dnv = sg.in(n_veh, S); int32_t* dof = sg.in(old_flag, S);
    double* dX = sg.out(X, n_X); double* dU = sg.out(U, n_U); double* dc = sg.out(cost, B); double* dk = sg.out(kkt, B);
    int32_t* ds = sg.out(status, B); int32_t* di = sg.out(iters, B);
    int32_t* dfl = sg.out(flag, S); double* dsc = sg.out(sel_cost, S * R); double* dbX = sg.out(best_X, n_bX);
    if (int rc = sg.up(g_stream)) return rc;
    if (int rc = crx_planner_plan_dev(d, sd, n_scen, dx0, dbs, dbe, dlb, dub, dnv, dos, doe, dof, dX, dU, dc, ds, dk, di,
                                      dfl, dsc, dbX, g_stream)) return rc;
    return sg.down(g_stream);
}

}  // extern "C"

// ---- dispatch order of a solver launch ------------------------------------------------------------------------------
extern "C" {

int crx_order_longest_first_dev(int batch, const int32_t* iters, const int32_t* active, int32_t* order, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (batch < 0) return fail(CRX_ERR_ARG, "bad batch");
    if (batch == 0) return CRX_OK;
    if (!iters || !order) return fail(CRX_ERR_ARG, "NULL array argument");
    crx_order_kparams op;
    memset(&op, 0, sizeof(op));
    op.batch = batch; op.mode = 0; op.iters = iters; op.active = active; op.order = order;
    hipError_t e = crx_launch_order(op, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "order launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_cbf_order_dev(const crx_cbf_desc* d, int batch, const int32_t* active, const double* x0, const double* xt,
                      const double* obs_s, const double* obs_ey, const double* lap_off, const int32_t* n_obs, const double* obs_dims,
                      int32_t* order, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (!d || batch < 0) return fail(CRX_ERR_ARG, "bad descriptor / batch");
    if (d->N < 1 || d->N > CRX_MAX_N || d->n_obs_max < 0 || d->n_obs_max > CRX_MAX_OBS) return fail(CRX_ERR_ARG, "bad N / n_obs_max");
    if (d->degree != 2 && d->degree != 4 && d->degree != 6 && d->degree != 8) return fail(CRX_ERR_ARG, "degree must be 2, 4, 6 or 8");
    if (batch == 0) return CRX_OK;
    if (!x0 || !xt || !order || (d->n_obs_max > 0 && (!obs_s || !obs_ey || !lap_off || !n_obs))) return fail(CRX_ERR_ARG, "NULL array argument");
    crx_order_kparams op;
    memset(&op, 0, sizeof(op));
    op.batch = batch; op.mode = 1; op.active = active; op.order = order;
    op.V = d->n_obs_max; op.stride = d->N + 1; op.degree = d->degree; op.margin = d->margin; op.l_sum = d->l_sum; op.w_sum = d->w_sum;
    op.per_stage_target = d->per_stage_target; op.ds_per_vx = d->A[4 * 6 + 0]; op.xt = xt;
    op.x0 = x0; op.obs_s = obs_s; op.obs_ey = obs_ey; op.lap_off = lap_off; op.obs_dims = d->n_obs_max > 0 ? obs_dims : nullptr; op.n_obs = n_obs;
    hipError_t e = crx_launch_order(op, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "order launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

}  // extern "C"

// ---- device-resident racing-game loop: bookkeeping between the solver launches ----------------------------------
extern "C" {

int crx_game_traffic_dev(int N, int batch, int n_cars, double lap_length, double t, double dt, const double* car_s0, const double* car_v,
                         const double* car_ey, double* veh_xcurv, double* pred_s, double* pred_ey, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (N < 1 || N > CRX_MAX_N || batch < 0 || n_cars < 0 || !(lap_length > 0.0)) return fail(CRX_ERR_ARG, "bad game traffic dimensions");
    if (batch == 0 || n_cars == 0) return CRX_OK;
    if (!car_s0 || !car_v || !car_ey || !veh_xcurv || !pred_s || !pred_ey) return fail(CRX_ERR_ARG, "NULL array argument");
    crx_game_kparams gp;
    memset(&gp, 0, sizeof(gp));
    gp.Np = N; gp.batch = batch; gp.n_cars = n_cars; gp.lap_length = lap_length; gp.t = t; gp.dt = dt;
    gp.car_s0 = car_s0; gp.car_v = car_v; gp.car_ey = car_ey; gp.veh_xcurv = veh_xcurv; gp.pred_s = pred_s; gp.pred_ey = pred_ey;
    hipError_t e = crx_launch_game(0, gp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "game traffic launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_game_masks_dev(int batch, const int32_t* n_veh, const int32_t* overflow, int32_t* m_overtake, int32_t* m_lmpc,
                       int32_t* overflow_seen, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (batch < 0) return fail(CRX_ERR_ARG, "batch < 0");
    if (batch == 0) return CRX_OK;
    if (!n_veh || !m_overtake || !m_lmpc) return fail(CRX_ERR_ARG, "NULL array argument");
    crx_game_kparams gp;
    memset(&gp, 0, sizeof(gp));
    gp.batch = batch; gp.n_veh = n_veh; gp.m_overtake = m_overtake; gp.m_lmpc = m_lmpc;
    if (overflow && overflow_seen) { gp.overflow = overflow; gp.overflow_seen = overflow_seen; }
    hipError_t e = crx_launch_game(1, gp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "game masks launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_game_commit_dev(int N, int Np, int batch, const int32_t* overtake, const double* U_track, const double* X_lmpc,
                        const double* U_lmpc, const int32_t* flag, double* u, double* u_old, double* u_prev, double* lin_points,
                        double* lin_input, int32_t* step_no, int32_t* addpoint_step, int32_t* old_flag, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (N < 1 || N > CRX_MAX_N || batch < 0 || (overtake && (Np < 1 || Np > CRX_MAX_N))) return fail(CRX_ERR_ARG, "bad game commit dimensions");
    if (batch == 0) return CRX_OK;
    if (!X_lmpc || !U_lmpc || !u || !u_old || !u_prev || !lin_points || !lin_input || !step_no || !addpoint_step)
        return fail(CRX_ERR_ARG, "NULL array argument");
    if (overtake && (!U_track || (old_flag && !flag))) return fail(CRX_ERR_ARG, "NULL overtake-branch array");
    crx_game_kparams gp;
    memset(&gp, 0, sizeof(gp));
    gp.N = N; gp.Np = Np; gp.batch = batch; gp.overtake = overtake; gp.U_track = U_track; gp.X_lmpc = X_lmpc; gp.U_lmpc = U_lmpc;
    gp.flag = flag; gp.u = u; gp.u_old = u_old; gp.u_prev = u_prev; gp.lin_points = lin_points; gp.lin_input = lin_input;
    gp.step_no = step_no; gp.addpoint_step = addpoint_step; gp.old_flag = old_flag;
    hipError_t e = crx_launch_game(2, gp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "game commit launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

int crx_game_log_dev(int batch, int n_points, double lap_length, const double* xcurv, const double* u, const int32_t* laps,
                     int32_t* laps_prev, double* log_x, double* log_u, int32_t* n_log, int32_t* crossed, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (batch < 0 || n_points < 2) return fail(CRX_ERR_ARG, "bad game log dimensions");
    if (batch == 0) return CRX_OK;
    if (!xcurv || !u || !laps || !laps_prev || !log_x || !log_u || !n_log || !crossed) return fail(CRX_ERR_ARG, "NULL array argument");
    crx_game_kparams gp;
    memset(&gp, 0, sizeof(gp));
    gp.batch = batch; gp.n_points = n_points; gp.lap_length = lap_length; gp.xcurv = xcurv; gp.u = (double*)u; gp.laps = laps;
    gp.laps_prev = laps_prev; gp.log_x = log_x; gp.log_u = log_u; gp.n_log = n_log; gp.crossed = crossed;
    hipError_t e = crx_launch_game(3, gp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CRX_ERR_HIP, "game log launch: %s", hipGetErrorString(e));
    return CRX_OK;
}

}  // extern "C"

// ---- multi-GPU: the ONE collective of the planner sweep, on RCCL directly ----------------------------------------
// One process per GPU; the caller owns the rendezvous (it carries the 128-byte id from rank 0 to the others by whatever
// it has: MPI, a file, torch.distributed's store).  RCCL is resolved at run time: if the process already holds a librccl
// (PyTorch-ROCm bundles one under the SONAME librccl.so.1) that copy is used -- two RCCL instances in one process would
// each build their own topology and IPC state -- otherwise /opt/rocm's is loaded.  libcrx therefore has no link-time
// dependency on RCCL, and single-GPU users never load it.
namespace {
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;
ncclComm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;

int rccl_bind() {
    if (g_rccl.AllGather) return CRX_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;     // a copy the process already holds
    if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail(CRX_ERR_HIP, "RCCL not found (librccl.so.1): %s", dlerror());
    g_rccl.h = h;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    auto ag = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.GetErrorString || !ag)
        return fail(CRX_ERR_HIP, "librccl lacks an expected entry point");
    g_rccl.AllGather = ag;
    return CRX_OK;
}
#define RCCL_TRY(expr)                                                                                    \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) return fail(CRX_ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString(r_));      \
    } while (0)

// winner record = {int32 flag; int32 status; double X[N+1][6]}: 8 + 624 = 632 B at N = 12, carried as 1 + 6 (N + 1)
// 8-byte words (word 0 = the two int32, bit for bit; the collective only moves bytes); rows past n_local are zero
__global__ void __launch_bounds__(256) crx_pack_winners_kernel(int n_local, int n_max, int rec, const int32_t* flag, const int32_t* status,
                                                               const double* best_X, double* send) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_max * rec) return;
    const size_t s = i / rec;
    const int c = (int)(i - s * rec);
    double v = 0.0;
    if (s < (size_t)n_local) {
        if (c == 0) {
            const unsigned long long w = (unsigned long long)(uint32_t)flag[s] | ((unsigned long long)(uint32_t)(status ? status[s] : 0) << 32);
            v = __longlong_as_double((long long)w);
        } else {
            v = best_X[s * (rec - 1) + (c - 1)];
        }
    }
    send[i] = v;
}
}  // namespace

extern "C" {

int crx_comm_get_unique_id(void* id) {
    if (!id) return fail(CRX_ERR_ARG, "id is NULL");
    if (int rc = ensure_init()) return rc;
    if (int rc = rccl_bind()) return rc;
    static_assert(sizeof(ncclUniqueId) == CRX_COMM_ID_BYTES, "id size");
    RCCL_TRY(g_rccl.GetUniqueId((ncclUniqueId*)id));
    return CRX_OK;
}

int crx_comm_init_rank(const void* id, int world, int rank) {
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(CRX_ERR_ARG, "bad communicator arguments (world %d, rank %d)", world, rank);
    if (int rc = ensure_init()) return rc;
    if (int rc = rccl_bind()) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_comm) return fail(CRX_ERR_ARG, "communicator already initialised (crx_comm_destroy first)");
    HIP_TRY(hipSetDevice(g_device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    RCCL_TRY(g_rccl.CommInitRank(&g_comm, world, uid, rank));
    g_world = world; g_rank = rank;
    return CRX_OK;
}

// one wavefront that keeps a CU busy for `ticks` of the constant 100 MHz clock: two of them on two streams finish in the time of
// one if, and only if, the streams sit on different hardware queues
__global__ void __launch_bounds__(64) crx_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static double spin_ms(hipStream_t a, hipStream_t b, long long ticks) {
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(crx_spin_kernel, dim3(1), dim3(64), 0, a, ticks);
    if (b) hipLaunchKernelGGL(crx_spin_kernel, dim3(1), dim3(64), 0, b, ticks);
    (void)hipStreamSynchronize(a);
    if (b) (void)hipStreamSynchronize(b);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int crx_streams_create(int n, void** streams, int* n_concurrent) {
    if (int rc = ensure_init()) return rc;
    if (n < 0 || n > 64 || (n > 0 && !streams)) return fail(CRX_ERR_ARG, "bad stream count / NULL array");
    if (n_concurrent) *n_concurrent = 0;
    if (n == 0) return CRX_OK;
    HIP_TRY(hipSetDevice(g_device));
    // candidates: more than asked for, so that a set of n on pairwise different queues can be picked (if the runtime has that many)
    const int m = n + 8;
    std::vector<hipStream_t> cand(m, nullptr);
    for (int i = 0; i < m; i++) {
        hipError_t e = hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking);
        if (e != hipSuccess) {
            for (int j2 = 0; j2 < i; j2++) (void)hipStreamDestroy(cand[j2]);
            return fail(CRX_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(e));
        }
    }
    const long long ticks = 15000;   // 150 us
    for (int i = 0; i < m; i++) (void)spin_ms(cand[i], nullptr, 100);   // first use: the runtime binds the queue, the kernel is loaded
    double single = 1e30;
    for (int i = 0; i < m; i++) single = fmin(single, spin_ms(cand[i], nullptr, ticks));
    // greedy: a stream joins the set if a pair of spins with every member takes the time of one (twice measured, the faster counts)
    std::vector<int> pick;
    for (int i = 0; i < m && (int)pick.size() < n; i++) {
        bool ok = true;
        for (int p : pick) {
            const double t = fmin(spin_ms(cand[p], cand[i], ticks), spin_ms(cand[p], cand[i], ticks));
            if (t > 1.6 * single) { ok = false; break; }
        }
        if (ok) pick.push_back(i);
    }
    const int nc = (int)pick.size();
    std::vector<char> used(m, 0);
    for (int p : pick) used[p] = 1;
    for (int i = 0; i < m && (int)pick.size() < n; i++)      // fewer queues than streams asked for: the rest share
        if (!used[i]) { pick.push_back(i); used[i] = 1; }
    for (int i = 0; i < n; i++) streams[i] = (void*)cand[pick[i]];
    for (int i = 0; i < m; i++)
        if (!used[i]) (void)hipStreamDestroy(cand[i]);
    if (n_concurrent) *n_concurrent = nc;
    return CRX_OK;
}

int crx_streams_destroy(int n, void** streams) {
    if (n < 0 || (n > 0 && !streams)) return fail(CRX_ERR_ARG, "bad stream count / NULL array");
    int rc = CRX_OK;
    for (int i = 0; i < n; i++) {
        if (!streams[i]) continue;
        (void)hipStreamSynchronize((hipStream_t)streams[i]);
        if (hipStreamDestroy((hipStream_t)streams[i]) != hipSuccess) rc = fail(CRX_ERR_HIP, "hipStreamDestroy failed");
        streams[i] = nullptr;
    }
    return rc;
}

int crx_comm_world(void) { return g_comm ? g_world : 0; }
int crx_comm_rank(void) { return g_comm ? g_rank : -1; }

int crx_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return CRX_OK;
    RCCL_TRY(g_rccl.CommDestroy(g_comm));
    g_comm = nullptr; g_world = 0; g_rank = -1;
    return CRX_OK;
}

int crx_allgather_winners_dev(int n_local, int n_max, int N, const int32_t* flag, const int32_t* status, const double* best_X, double* send,
                              double* recv, void* stream) {
    if (int rc = ensure_init()) return rc;
    if (!g_comm) return fail(CRX_ERR_ARG, "no communicator (crx_comm_init_rank)");
    if (n_local < 0 || n_max < n_local || N < 1 || N > CRX_MAX_N) return fail(CRX_ERR_ARG, "bad sizes (n_local %d, n_max %d, N %d)", n_local, n_max, N);
    if (n_max == 0) return CRX_OK;
    if (!send || !recv || (n_local > 0 && (!flag || !best_X))) return fail(CRX_ERR_ARG, "NULL array argument");
    const int rec = 1 + (N + 1) * 6;
    const size_t cnt = (size_t)n_max * rec;
    hipLaunchKernelGGL(crx_pack_winners_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_local, n_max, rec, flag,
                       status, best_X, send);
    HIP_TRY(hipGetLastError());
    RCCL_TRY(g_rccl.AllGather(send, recv, cnt, ncclFloat64, g_comm, (hipStream_t)stream));
    return CRX_OK;
}

}  // extern "C"
