// nnlm_mi355x.hip -- host side of libnnlm_mi355x.so: device-resident state (nnlm_handle), kernel
// launches for one half-step, the alternating driver restating c_nnmf / c_nnlm, and the C ABI of
// include/nnlm_mi355x.h.  gfx950 only; no CPU fallback: every entry fails with NNLM_ERR_HIP when no
// device is present.
//
// Reference map (relative to /root/reference):
//   nnlm_c_nnmf      <- c_nnmf            src/nnmf.cpp:4-220
//   nnlm_c_nnlm      <- c_nnlm            src/nnlm.cpp:4-53
//   half_step()      <- update()          src/update_with_missing.cpp:3-55
//                       update_with_missing()  src/update_with_missing.cpp:58-139
//   penalties()      <- add_penalty()     src/nnmf.cpp:224-240
#include "../../include/nnlm_mi355x.h"
#include "common.h"
#include "k_errors.h"
#include "k_generic.h"
#include "k_gram.h"
#include "k_kl.h"
#include "k_missing.h"
#include "k_prep.h"
#include "k_sweep.h"
#include "k_sweep_q.h"
#include "k_xprod.h"
#include "k_xprod16.h"
#include "tu_sweepq.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

static thread_local std::string g_last_error;

enum ProfId { P_XPROD_H = 0, P_XPROD_W, P_GRAM, P_SWEEP_H, P_SWEEP_W, P_ERRORS, P_XPROD_W_ERR, P_ALLGATHER, P_ALLREDUCE, P_UNPACK, P_ERR_REDUCE, P_COUNT };
// ("xprod_w_err": W half-step cross products that also evaluate the error sums -- the fused launches have a scope of their own;
//  "allgather" / "allreduce": the RCCL collective of a sharded half-step between two events on the stream it is enqueued on;
//  "unpack": shard_unpack_kernel + the sum of the ranks' Gram partial sums behind it)
//  "errors": a separate pass over A (errors_f32_kernel / errors64_kernel + its reduction); "err_reduce": the reduction of the partial sums the
//  fused cross product left behind -- 5 us, no HBM work: it must not be priced as a pass over A
static const char *kProfNames[P_COUNT] = {"xprod_h", "xprod_w", "gram", "sweep_h", "sweep_w", "errors", "xprod_w_err", "allgather", "allreduce", "unpack", "err_reduce"};

struct ProfRec {
    int id;
    hipEvent_t e0, e1;
};

struct nnlm_handle {
    int device = 0;
    int cus = 256;              // compute units of the device (sweep launch policy: one wavefront of the SCD sweep per SIMD, 4 SIMDs per CU)
    int sweep_wgs = 0;          // workgroups of the last sweep_scd_q(w)_kernel launch = Gram partial-sum slabs it left behind
    int cur_which = 1;          // the half-step in progress (0: W, 1: H)
    int kl_form[2] = {-1, -1}; // per half-step: its last KL solver launch (0 tile kernel on the GEMM's states, 1 tile kernel on its own states, 2 strict register kernel, 3 streaming)
    int sweep_form[2] = {-1, -1}, sweep_groups[2] = {0, 0}; // per half-step (0: W, 1: H): form of its last SCD sweep launch (0 plain, 1 persistent) and
                                                             // column groups per workgroup (nnlm_get_info)
    int prec = NNLM_PREC_F32;
    hipStream_t stream = nullptr;   // main: cross products, solvers
    hipStream_t stream_e = nullptr; // error block, concurrent with the (speculative) next W half-step
    hipEvent_t ev_hdone = nullptr, ev_err = nullptr, ev_xdone = nullptr;
    std::string err;

    // problem
    int n = 0, m = 0, npad = 0, mpad = 0;
    void *A = nullptr;          // T [mpad][npad]
    uint32_t *miss = nullptr;   // [mpad][npad/32]
    uint32_t *missT = nullptr;  // [npad][mpad/32], only when any_missing
    bool any_missing = false;
    bool dense_cols = true;     // multi-GPU form of the dense square-loss half-step (half_step): column-sharded (true) or all-reduce
    int upk_max_for = -1;       // multi-GPU: the half-step (0: W, 1: H) whose unpack left max|factor| in maxbits[6 + which], or -1
    int gshard_for = -1;        // multi-GPU: the factor (0: W, 1: H) whose Gram the last unpack summed into Graw from the ranks' partial sums, or -1
    int y16_for = -1;           // multi-GPU: the factor whose split-fp16 copy (and exponent, scal_exp[1]) the last unpack left in Y16, or -1
    size_t pack_tail = 0;       // doubles behind the k x cpr slab in the packed payload of the current half-step (KP * KP Gram partial sums, or 0)
    unsigned *fixed_maxw = nullptr; // word holding max|fixed factor| of the half-step in progress (split-fp16 copies)
    double n_non_missing = 0.0, kl_const = 0.0;

    // factors
    int k = 0, NKQ = 0, KP = 0, KP8 = 0;
    double *W64 = nullptr, *H64 = nullptr;        // [KP][npad], [KP][mpad]  (W64 = W64b[wcur])
    void *Wop = nullptr;                          // T [KP][npad] (aliases W64 in f64 mode)  (Wop = Wopb[wcur])
    double *W64b[2] = {nullptr, nullptr};         // W is double buffered: every W half-step writes the other buffer, so a
    void *Wopb[2] = {nullptr, nullptr};           // speculative half-step can be dropped and the error block can read W_i
    int wcur = 0;
    float *Hkq = nullptr;                         // fp32 [KP][mpad] copy of H, refreshed by nnlm_errors (f32 mode)
    unsigned long long *Wmask = nullptr, *Hmask = nullptr; // [npad][MW], [mpad][MW]: bit q of a column's words = entry (q, col) is masked
    bool has_wmask = false, has_hmask = false;
    int MW = 1;                                   // 64-bit mask words per column = ceil(k / 64)
    void *AT = nullptr;                           // T [npad][mpad]: contraction-contiguous copy of A for the W half-step of the paths that
                                                  // stream it with the TN kernels (rank > 64) or per column (KL, F32 mode); made on first use
    double *klsw = nullptr, *klsw_cols = nullptr; // KL: row sums of the fixed factor [KP] / per column over its non-missing entries [cols][KP]
    void *klst = nullptr;                         // kl_stream_kernel: [cols][2][ld] state vectors + data columns
    size_t klst_bytes = 0;
    bool what_tight = false;                      // the matrix-sized What / What64 could not be had for this matrix (not retried every half-step)
    bool klst_tight = false;                      // the full-size streaming scratch could not be had: the chunked one is kept
    bool at_tight = false;                        // the transposed copy AT could not be had (soft request of the KL solvers)

    // workspaces
    double *Cx = nullptr;
    size_t Cx_elems = 0;
    double *gslabs = nullptr, *Graw = nullptr; // Graw = head of red
    double *red = nullptr;               // [KP*KP | KP*max(npad,mpad)]: the buffer one all-reduce sums
    // NA path: per orientation (0: rows of A for the W half-step, 1: columns for the H half-step) the CSR lists of the rows a
    // column's Gram sums over (k_missing.h); built on first use, they live as long as the matrix
    uint32_t *na_ptr[2] = {nullptr, nullptr}, *na_meta[2] = {nullptr, nullptr};
    int *na_idx[2] = {nullptr, nullptr};
    double *Yrow = nullptr;              // [max(npad,mpad)][KP] row-major copy of the fixed factor (NA path)
    double *Gcols = nullptr;             // [max(n,m)][KP][KP] per-column Grams (NA path)
    double *partials = nullptr;
    size_t partials_elems = 0;
    double *scal = nullptr;              // 16 doubles of reduction results
    unsigned long long *sweeps = nullptr; // [2] device counters; sw_active = the one the current trace window sums into
    int sw_active = 0;
    double *host_res = nullptr;           // pinned: 8 reduction results + sweep counter of the asynchronous error block

    // multi-GPU
    int rank = 0, nranks = 1;
    void *comm = nullptr;
    bool sharded = false;       // nranks > 1, or a real 1-rank communicator (exercises the sharded code path on one GPU)
    double *pack_send = nullptr; // [KP][cpr]: this rank's updated columns, contiguous for ncclAllGather
    double *pack_all = nullptr;  // [nranks][KP][cpr]
    size_t pack_elems = 0;
    double *sweepq_img = nullptr;             // operand image of sweep_scd_q_kernel (k_sweep_q.h), rewritten every half-step
    // split-fp16 cross products (k_xprod16.h; F32 mode, single GPU): A16 [mpad][npad], A16T [npad][mpad], Y16 [KP][max(npad,mpad)]
    bool x16 = false;
    uint32_t *A16 = nullptr, *A16T = nullptr, *Y16 = nullptr;
    unsigned *maxbits = nullptr; // device [16]: bit patterns of max|factor|: [0] absmax_f64_kernel, [1],[2] alternately gram_partial_kernel, [3] block counter, [6],[7] shard_unpack_kernel (W, H),
                                 // [4],[5] alternately the fast sweep kernel's own max of what it solved (k_sweep_q.h), [8] the same for a rank's
                                 // column shard (travels behind its packed slab: gram_fold_tail_kernel clears it)
    // What the fast sweep leaves behind for the next half-step (dense one-GPU split-fp16 path): max|x| in maxbits[4 + sg_par] and
    // sg_nslabs Gram partial sums (one per workgroup) in sg_slabs.  sg_which: the factor they describe (1 = H, 0 = W, -1 = none).
    int sg_which = -1, sg_par = 0, sg_nslabs = 0;
    int sg_other = -1;           // the factor whose max the OTHER word (maxbits[4 + (sg_par ^ 1)]) still holds, or -1
    int sg_prev = -1;            // sg_which as the current half-step found it (becomes sg_other after its sweep)
    bool sg_request = false;     // set by half_step for the sweep it is about to launch
    double *sg_slabs = nullptr;
    int mb_par = 0;
    bool pack_ready = false;     // sweepq_img already produced for this half-step (the one-stream dense flow packs it ahead of the cross product)
    int *scal_exp = nullptr;     // device: {eA, eY, eW of the fused error block}
    float *What = nullptr;       // [mpad][npad] fp32 W^T H: starting state vectors of a KL half-step (wh_store_kernel), on first use
    double *What64 = nullptr;    // the same in fp64 for the strict mode (wh_store64_kernel)
    uint32_t *W16c = nullptr, *H16c = nullptr; // kq-contiguous split copies [npad][2][64], [mpad][2][64] (fused error block)
    bool fuse_err = false;       // request: the next W half-step's cross product also evaluates the error sums of (W, H) now current
    int cus_device = 0;          // the device's CU count (cus may be the test hook's)
    int fused_nb = 0;            // answer: number of (sum of squares, KL) pairs it left in `partials` (0 = not fused)
    // multi-GPU: the error block of a trace iteration is enqueued from INSIDE the speculative W half-step, right behind its cross product:
    // its all-reduces (two sums, one sweep counter) must be issued ahead of that half-step's all-gather -- RCCL runs a communicator's
    // collectives in issue order, and behind the all-gather the host would learn the stopping decision only when the whole half-step is done
    bool err_hook = false, err_need_pen = true, err_launched = false;
    unsigned *err_zero_word = nullptr; // the fused kernel clears this word (max|x| of the sweep that follows it; see factor16_fold_err_kernel)
    unsigned long long *sweeps_tmp = nullptr; // device scratch for the all-reduced sweep counter

    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    double prof_ms[P_COUNT] = {0};
    long long prof_n[P_COUNT] = {0};
};

// Dynamic LDS beyond 64 KB must be granted per kernel.  The launch helpers are void: a refusal is remembered here and reported by the
// next LAUNCHCHK (every half-step / error block ends with one), together with whatever the launch itself then raised.  The latch is
// per thread, not per handle: fail() and every public entry that launches clear it, so a refusal never outlives the call it
// happened in and is never reported against another handle's launch.
static thread_local hipError_t g_attr_err = hipSuccess;
static thread_local const char *g_attr_what = "";

static int fail(nnlm_handle *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (h) h->err = buf;
    g_attr_err = hipSuccess; // (a refused attribute latched on the way here belongs to this failure, not to a later launch)
    return code;
}

#define HIPCHK(h, call)                                                                                              \
    do {                                                                                                             \
        hipError_t e__ = (call);                                                                                     \
        if (e__ != hipSuccess) return fail(h, NNLM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

static inline void set_dyn_lds(const void *fn, int lds, const char *what)
{
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess && g_attr_err == hipSuccess) {
        g_attr_err = e;
        g_attr_what = what;
    }
}
#define LAUNCHCHK(h)                                                                                                              \
    do {                                                                                                                          \
        if (g_attr_err != hipSuccess) {                                                                                           \
            const hipError_t ea__ = g_attr_err;                                                                                   \
            g_attr_err = hipSuccess;                                                                                              \
            (void)hipGetLastError();                                                                                              \
            return fail(h, NNLM_ERR_HIP, "hipFuncSetAttribute(%s, MaxDynamicSharedMemorySize) failed: %s", g_attr_what, hipGetErrorString(ea__)); \
        }                                                                                                                         \
        HIPCHK(h, hipGetLastError());                                                                                             \
    } while (0)

// split-fp16 cross products (k_xprod16.h): THE cross products of the F32 mode
static bool x16_enabled(int precision) { return precision == NNLM_PREC_F32; }

static size_t esize(const nnlm_handle *h) { return h->prec == NNLM_PREC_F64 ? 8 : 4; }

// ---------------------------------------------------------------------------------------------
// profiling helpers
// ---------------------------------------------------------------------------------------------
struct ProfScope {
    nnlm_handle *h;
    ProfRec r;
    bool on;
    hipStream_t st;
    ProfScope(nnlm_handle *h_, int id, hipStream_t st_ = nullptr) : h(h_), on(h_->prof), st(st_ ? st_ : h_->stream)
    {
        if (on) {
            r.id = id;
            hipEventCreate(&r.e0);
            hipEventCreate(&r.e1);
            hipEventRecord(r.e0, st);
        }
    }
    ~ProfScope()
    {
        if (on) {
            hipEventRecord(r.e1, st);
            h->recs.push_back(r);
        }
    }
};

static void sync_all(nnlm_handle *h)
{
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->stream_e) hipStreamSynchronize(h->stream_e);
}

static void prof_collect(nnlm_handle *h)
{
    if (h->recs.empty()) return;
    sync_all(h);
    for (auto &r : h->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            h->prof_ms[r.id] += ms;
            h->prof_n[r.id] += 1;
        }
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
    }
    h->recs.clear();
}

// RCCL entry points, resolved lazily with dlopen (single-GPU runs never touch librccl)
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

static int rccl_load()
{
    if (g_rccl.lib) return NNLM_OK;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(nullptr, NNLM_ERR_COMM, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(lib, "ncclAllGather");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return fail(nullptr, NNLM_ERR_COMM, "librccl lacks a required symbol");
    g_rccl.lib = lib;
    return NNLM_OK;
}

// ---------------------------------------------------------------------------------------------
// create / destroy
// ---------------------------------------------------------------------------------------------
extern "C" int nnlm_abi_version(void) { return NNLM_ABI_VERSION; }

extern "C" const char *nnlm_last_error(const nnlm_handle *h) { return h ? h->err.c_str() : g_last_error.c_str(); }

extern "C" unsigned nnlm_trace_capacity(unsigned max_iter, unsigned trace)
{
    if (trace < 1) trace = 1;
    return (unsigned)std::ceil((double)max_iter / (double)trace) + 1; // src/nnmf.cpp:53-54
}

static std::atomic<int> g_debug_cus{0}; // nnlm_debug_set_cus (test hook, read once per nnlm_create): compute units the launch policies of new handles count (0 = the device's)
// Pinned bounce buffers of nnlm_set_matrix, kept between calls.  Pinning costs ~0.3 ms per MB: two fresh 64 MB buffers were ~40 ms of EVERY
// upload (half of config 2's 78 ms, most of the 53 ms an 80 MB matrix took -- scripts/gpu_call_breakdown.py).  One pair per process, taken
// by the call that finds it free (a concurrent upload on another thread allocates its own and frees it), released at exit.
static std::mutex g_bounce_mu;
static double *g_bounce[2] = {nullptr, nullptr};
static size_t g_bounce_bytes = 0;
static bool g_bounce_busy = false, g_bounce_atexit = false;
static void bounce_release_at_exit()
{
    // (the HIP runtime may already be gone when exit handlers run: errors are ignored)
    for (int b = 0; b < 2; b++)
        if (g_bounce[b]) (void)hipHostFree(g_bounce[b]);
    g_bounce[0] = g_bounce[1] = nullptr;
    g_bounce_bytes = 0;
}
// both buffers of at least `bytes`, or false (none to be had / the pair is in use): *cached tells the caller whether to give them back
static bool bounce_acquire(size_t bytes, double *out[2], bool *cached)
{
    {
        std::lock_guard<std::mutex> lk(g_bounce_mu);
        if (!g_bounce_busy) {
            if (g_bounce_bytes < bytes) {
                for (int b = 0; b < 2; b++)
                    if (g_bounce[b]) (void)hipHostFree(g_bounce[b]);
                g_bounce[0] = g_bounce[1] = nullptr;
                g_bounce_bytes = 0;
                if (hipHostMalloc(&g_bounce[0], bytes) != hipSuccess || hipHostMalloc(&g_bounce[1], bytes) != hipSuccess) {
                    (void)hipGetLastError();
                    for (int b = 0; b < 2; b++)
                        if (g_bounce[b]) (void)hipHostFree(g_bounce[b]);
                    g_bounce[0] = g_bounce[1] = nullptr;
                    return false;
                }
                g_bounce_bytes = bytes;
                if (!g_bounce_atexit) {
                    g_bounce_atexit = true;
                    atexit(bounce_release_at_exit);
                }
            }
            g_bounce_busy = true;
            out[0] = g_bounce[0], out[1] = g_bounce[1];
            *cached = true;
            return true;
        }
    }
    *cached = false; // another upload holds the pair: buffers of this call's own
    out[0] = out[1] = nullptr;
    if (hipHostMalloc(&out[0], bytes) != hipSuccess || hipHostMalloc(&out[1], bytes) != hipSuccess) {
        (void)hipGetLastError();
        for (int b = 0; b < 2; b++)
            if (out[b]) (void)hipHostFree(out[b]);
        out[0] = out[1] = nullptr;
        return false;
    }
    return true;
}
static void bounce_give_back(double *buf[2], bool cached)
{
    if (cached) {
        std::lock_guard<std::mutex> lk(g_bounce_mu);
        g_bounce_busy = false;
    } else
        for (int b = 0; b < 2; b++)
            if (buf[b]) (void)hipHostFree(buf[b]);
    buf[0] = buf[1] = nullptr;
}

static std::atomic<size_t> g_debug_alloc_limit{0}; // nnlm_debug_alloc_limit (test hook): matrix-sized KL workspaces beyond this many bytes "do not fit" (0 = no limit)

// Matrix-sized workspaces of the KL solvers (starting states of all columns, transposed copy of A, streaming scratch): the callers have
// a smaller-footprint path when one of them cannot be had, so a failure here is an answer, not an error.
static hipError_t big_malloc(void **p, size_t bytes)
{
    *p = nullptr;
    const size_t lim = g_debug_alloc_limit.load(std::memory_order_relaxed);
    if (lim && bytes > lim) return hipErrorOutOfMemory;
    const hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
    }
    return e;
}

extern "C" int nnlm_debug_alloc_limit(size_t bytes)
{
    g_debug_alloc_limit = bytes;
    return NNLM_OK;
}

extern "C" int nnlm_debug_set_cus(int cus)
{
    if (cus < 0) return fail(nullptr, NNLM_ERR_ARG, "nnlm_debug_set_cus: %d", cus);
    g_debug_cus = cus;
    return NNLM_OK;
}

// The streams, events and small buffers of a destroyed handle wait here for the next nnlm_create on the same device (one set per process):
// creating and destroying them is 5 + 5 ms of every one-shot call -- half of a 200 x 100 nnmf() of 100 iterations (scripts/gpu_call_breakdown.py).
// The handle itself is always a fresh object: no state of a previous call survives, only resources do.
struct HandleRes {
    bool valid = false;
    int device = -1, cus = 0;
    hipStream_t stream = nullptr, stream_e = nullptr;
    hipEvent_t ev_hdone = nullptr, ev_err = nullptr, ev_xdone = nullptr;
    double *scal = nullptr, *host_res = nullptr, *sweepq_img = nullptr;
    unsigned long long *sweeps = nullptr, *sweeps_tmp = nullptr;
    unsigned *maxbits = nullptr;
    int *scal_exp = nullptr;
};
static std::mutex g_res_mu;
static HandleRes g_res;
static bool g_res_atexit = false;
static void handle_res_destroy(HandleRes &r)
{
    (void)hipFree(r.scal), (void)hipFree(r.sweeps), (void)hipHostFree(r.host_res), (void)hipFree(r.sweeps_tmp), (void)hipFree(r.sweepq_img);
    (void)hipFree(r.maxbits), (void)hipFree(r.scal_exp);
    if (r.ev_hdone) (void)hipEventDestroy(r.ev_hdone);
    if (r.ev_err) (void)hipEventDestroy(r.ev_err);
    if (r.ev_xdone) (void)hipEventDestroy(r.ev_xdone);
    if (r.stream_e) (void)hipStreamDestroy(r.stream_e);
    if (r.stream) (void)hipStreamDestroy(r.stream);
    r = HandleRes{};
}
static void handle_res_release_at_exit()
{
    if (g_res.valid) handle_res_destroy(g_res); // (errors ignored: the runtime may be shutting down)
}

extern "C" int nnlm_create(nnlm_handle **out, int device, int precision)
{
    if (!out) return fail(nullptr, NNLM_ERR_ARG, "nnlm_create: out is NULL");
    *out = nullptr;
    if (precision != NNLM_PREC_F32 && precision != NNLM_PREC_F64) return fail(nullptr, NNLM_ERR_ARG, "nnlm_create: unknown precision %d", precision);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NNLM_ERR_HIP, "nnlm_create: no HIP device available (%s); libnnlm_mi355x has no CPU path", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return fail(nullptr, NNLM_ERR_ARG, "nnlm_create: device %d out of range (0..%d)", device, ndev - 1);
    HIPCHK(nullptr, hipSetDevice(device));
    HandleRes res;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        if (g_res.valid && g_res.device == device) {
            res = g_res;
            g_res = HandleRes{};
        }
    }
    if (res.valid) { // the resources of a handle this process destroyed earlier (same device: the architecture check has been made)
        nnlm_handle *h = new nnlm_handle();
        h->device = device;
        h->prec = precision;
        h->cus = h->cus_device = res.cus;
        if (const int dc = g_debug_cus.load(std::memory_order_relaxed); dc > 0) h->cus = dc;
        h->stream = res.stream, h->stream_e = res.stream_e;
        h->ev_hdone = res.ev_hdone, h->ev_err = res.ev_err, h->ev_xdone = res.ev_xdone;
        h->scal = res.scal, h->sweeps = res.sweeps, h->host_res = res.host_res, h->sweeps_tmp = res.sweeps_tmp, h->sweepq_img = res.sweepq_img;
        h->maxbits = res.maxbits, h->scal_exp = res.scal_exp;
        // (inherited streams may be dead -- hipDeviceReset by the host application, a fork, a sticky error raised after the destroy: a
        //  cached set that does not answer is dropped and the call goes on to create a fresh one)
        const bool alive = hipMemsetAsync(h->sweeps, 0, 2 * sizeof(unsigned long long), h->stream) == hipSuccess &&
                           hipMemsetAsync(h->scal_exp, 0, 4 * sizeof(int), h->stream) == hipSuccess &&
                           hipMemsetAsync(h->maxbits, 0, 16 * sizeof(unsigned), h->stream) == hipSuccess &&
                           hipStreamSynchronize(h->stream) == hipSuccess && hipStreamSynchronize(h->stream_e) == hipSuccess;
        if (alive) {
            h->x16 = x16_enabled(precision);
            *out = h;
            return NNLM_OK;
        }
        (void)hipGetLastError();
        delete h; // (a plain object at this point: the resources are still `res`'s)
        handle_res_destroy(res);
    }
    hipDeviceProp_t prop;
    HIPCHK(nullptr, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, NNLM_ERR_HIP, "nnlm_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    nnlm_handle *h = new nnlm_handle();
    h->device = device;
    h->prec = precision;
    h->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->cus_device = h->cus;
    if (const int dc = g_debug_cus.load(std::memory_order_relaxed); dc > 0) h->cus = dc; // test hook (nnlm_debug_set_cus): the sweep's launch policy at small sizes
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream_e, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_hdone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_err, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_xdone, hipEventDisableTiming) != hipSuccess) {
        delete h;
        return fail(nullptr, NNLM_ERR_HIP, "nnlm_create: stream/event creation failed");
    }
    if (hipMalloc(&h->scal, 16 * sizeof(double)) != hipSuccess || hipMalloc(&h->sweeps, 2 * sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc(&h->host_res, 16 * sizeof(double)) != hipSuccess ||
        hipMalloc(&h->sweeps_tmp, sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&h->sweepq_img, sweepq_img_doubles(16, true) * sizeof(double)) != hipSuccess ||
        hipMalloc(&h->maxbits, 16 * sizeof(unsigned)) != hipSuccess || hipMalloc(&h->scal_exp, 4 * sizeof(int)) != hipSuccess) {
        delete h;
        return fail(nullptr, NNLM_ERR_HIP, "nnlm_create: hipMalloc failed");
    }
    if (hipMemsetAsync(h->sweeps, 0, 2 * sizeof(unsigned long long), h->stream) != hipSuccess ||
        hipMemsetAsync(h->scal_exp, 0, 4 * sizeof(int), h->stream) != hipSuccess ||
        hipMemsetAsync(h->maxbits, 0, 16 * sizeof(unsigned), h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
        const hipError_t em = hipGetLastError();
        HandleRes dead; // (not through nnlm_destroy: it would offer these resources to the next create)
        dead.stream = h->stream, dead.stream_e = h->stream_e, dead.ev_hdone = h->ev_hdone, dead.ev_err = h->ev_err, dead.ev_xdone = h->ev_xdone;
        dead.scal = h->scal, dead.sweeps = h->sweeps, dead.host_res = h->host_res, dead.sweeps_tmp = h->sweeps_tmp, dead.sweepq_img = h->sweepq_img;
        dead.maxbits = h->maxbits, dead.scal_exp = h->scal_exp;
        delete h;
        handle_res_destroy(dead);
        return fail(nullptr, NNLM_ERR_HIP, "nnlm_create: clearing the handle's counters failed: %s", hipGetErrorString(em));
    }
    h->x16 = x16_enabled(precision);
    *out = h;
    return NNLM_OK;
}

// The process-wide caches (the streams / events / scratch of the last destroyed handle, the pinned bounce buffers of nnlm_set_matrix: up to
// 2 x 64 MB of pinned host memory) are released at exit; an embedder that unloads the library earlier (an R package's .onUnload hook)
// or wants the pinned memory back calls this.  Handles in use are not affected.
extern "C" int nnlm_release_caches(void)
{
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        if (g_res.valid) handle_res_destroy(g_res);
    }
    {
        std::lock_guard<std::mutex> lk(g_bounce_mu);
        if (!g_bounce_busy) bounce_release_at_exit();
    }
    return NNLM_OK;
}

// Everything a half-step may find left behind by the previous one describes the factors as the library last wrote them: any other
// writer of the factors (nnlm_set_factors) or of the exchange (nnlm_comm_init) calls this.
static void invalidate_factor_caches(nnlm_handle *h)
{
    h->upk_max_for = -1;
    h->gshard_for = -1;
    h->y16_for = -1;
    h->sg_which = h->sg_other = -1;
    h->pack_ready = false;
}

static void free_factors(nnlm_handle *h)
{
    for (int i = 0; i < 2; i++) {
        if (h->Wopb[i] && h->Wopb[i] != (void *)h->W64b[i]) hipFree(h->Wopb[i]);
        hipFree(h->W64b[i]);
        h->W64b[i] = nullptr;
        h->Wopb[i] = nullptr;
    }
    h->wcur = 0;
    hipFree(h->H64);
    hipFree(h->Hkq);
    h->Hkq = nullptr;
    hipFree(h->Wmask);
    hipFree(h->Hmask);
    hipFree(h->Cx);
    hipFree(h->What);
    hipFree(h->What64);
    h->What = nullptr;
    h->What64 = nullptr;
    hipFree(h->klsw);
    hipFree(h->klsw_cols);
    hipFree(h->klst);
    h->klsw = h->klsw_cols = nullptr;
    h->klst = nullptr;
    h->klst_bytes = 0;
    h->what_tight = h->klst_tight = false; // (another rank, other workspace sizes: asked again)
    hipFree(h->Y16);
    hipFree(h->W16c);
    hipFree(h->H16c);
    h->Y16 = h->W16c = h->H16c = nullptr;
    hipFree(h->gslabs);
    hipFree(h->sg_slabs);
    h->sg_slabs = nullptr;
    h->sg_which = h->sg_other = -1;
    hipFree(h->red);
    hipFree(h->pack_send);
    hipFree(h->pack_all);
    h->pack_send = h->pack_all = nullptr;
    h->pack_elems = 0;
    hipFree(h->Yrow);
    hipFree(h->Gcols);
    h->red = h->Yrow = h->Gcols = nullptr;
    h->W64 = h->H64 = nullptr;
    h->Wop = nullptr;
    h->Wmask = h->Hmask = nullptr;
    h->Cx = h->gslabs = h->Graw = nullptr;
    h->k = 0;
}

static void free_matrix(nnlm_handle *h)
{
    for (int o = 0; o < 2; o++) {
        hipFree(h->na_ptr[o]);
        hipFree(h->na_meta[o]);
        hipFree(h->na_idx[o]);
        h->na_ptr[o] = h->na_meta[o] = nullptr;
        h->na_idx[o] = nullptr;
    }
    hipFree(h->A);
    hipFree(h->AT);
    h->AT = nullptr;
    h->what_tight = h->klst_tight = h->at_tight = false; // (what did not fit beside the last matrix may fit beside the next)
    hipFree(h->A16);
    hipFree(h->A16T);
    h->A16 = h->A16T = nullptr;
    hipFree(h->miss);
    hipFree(h->missT);
    hipFree(h->partials);
    h->A = nullptr;
    h->miss = nullptr;
    h->missT = nullptr;
    h->partials = nullptr;
    h->n = h->m = 0;
}

extern "C" void nnlm_destroy(nnlm_handle *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    sync_all(h);
    prof_collect(h);
    if (h->comm && g_rccl.CommDestroy) { // (before the streams its collectives were enqueued on go away)
        g_rccl.CommDestroy((ncclComm_t)h->comm);
        h->comm = nullptr;
    }
    free_factors(h);
    free_matrix(h);
    HandleRes res;
    res.valid = true, res.device = h->device, res.cus = h->cus_device;
    res.stream = h->stream, res.stream_e = h->stream_e;
    res.ev_hdone = h->ev_hdone, res.ev_err = h->ev_err, res.ev_xdone = h->ev_xdone;
    res.scal = h->scal, res.sweeps = h->sweeps, res.host_res = h->host_res, res.sweeps_tmp = h->sweeps_tmp, res.sweepq_img = h->sweepq_img;
    res.maxbits = h->maxbits, res.scal_exp = h->scal_exp;
    delete h;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        if (!g_res.valid && hipGetLastError() == hipSuccess) { // (the streams are idle: sync_all above; a sticky error keeps nothing)
            g_res = res;
            res.valid = false;
            if (!g_res_atexit) {
                g_res_atexit = true;
                atexit(handle_res_release_at_exit);
            }
        }
    }
    if (res.valid) handle_res_destroy(res);
}

// ---------------------------------------------------------------------------------------------
// matrix upload
// ---------------------------------------------------------------------------------------------
extern "C" int nnlm_set_matrix(nnlm_handle *h, const double *A, int n, int m)
{
    if (!h) return fail(nullptr, NNLM_ERR_ARG, "nnlm_set_matrix: handle is NULL");
    if (!A || n <= 0 || m <= 0) return fail(h, NNLM_ERR_ARG, "nnlm_set_matrix: A must be a non-empty n x m matrix (n=%d, m=%d)", n, m);
    HIPCHK(h, hipSetDevice(h->device));
    free_factors(h);
    free_matrix(h);
    h->n = n;
    h->m = m;
    h->npad = round_up_i(n, NNLM_PAD_N);
    h->mpad = round_up_i(m, NNLM_PAD_M);
    const size_t es = esize(h);
    const size_t a_bytes = (size_t)h->npad * h->mpad * es;
    const size_t miss_words = (size_t)h->mpad * (h->npad / 32);
    HIPCHK(h, hipMalloc(&h->A, a_bytes + 4096));
    HIPCHK(h, hipMalloc(&h->miss, miss_words * 4 + 64));
    HIPCHK(h, hipMemsetAsync(h->A, 0, a_bytes + 4096, h->stream));
    HIPCHK(h, hipMemsetAsync(h->miss, 0, miss_words * 4 + 64, h->stream));
    // partial-sum workspace shared by the prep / error / penalty reductions
    const size_t err_blocks = (size_t)(h->npad / ERR_TILE) * (h->mpad / ERR_TILE);
    const int gx = h->npad / PREP_BLOCK;
    const size_t prep_blocks = (size_t)gx * PREP_GRID_Y;
    h->partials_elems = 3 * (err_blocks > 2 * prep_blocks ? err_blocks : 2 * prep_blocks) + 64; // (two slots of prep partial sums)
    HIPCHK(h, hipMalloc(&h->partials, h->partials_elems * sizeof(double)));

    // Upload: the caller's matrix is pageable host memory (R's heap, a numpy array).  hipMemcpy from pageable memory stages through
    // ONE runtime thread (4.3 GB/s measured: 0.37 s of a 0.7 s nnmf() call of 500 iterations at config 2).  Here: chunks of whole
    // columns are copied by a few host threads into one of two PINNED bounce buffers, from there by DMA into one of two device
    // staging buffers, converted by prep_convert_kernel -- the host copy of chunk i + 1 runs beside the DMA and the prep pass of chunk i.
    const size_t stage_bytes_max = (size_t)64 << 20;
    int cols_per_chunk = (int)(stage_bytes_max / ((size_t)n * 8));
    if (cols_per_chunk < 1) cols_per_chunk = 1;
    if (cols_per_chunk > m) cols_per_chunk = m;
    const size_t chunk_bytes = (size_t)cols_per_chunk * n * 8;
    double *stage[2] = {nullptr, nullptr}, *bounce[2] = {nullptr, nullptr};
    double *hp[2] = {nullptr, nullptr}; // pinned: the prep pass's partial sums of each slot
    hipEvent_t ev_done[2] = {nullptr, nullptr};
    int rc = NNLM_OK;
    // (matrices of a few MB go through the runtime's pageable path: its own staging buffers are pinned already)
    bool bounce_cached = false;
    bool pinned = (size_t)n * m * 8 >= ((size_t)4 << 20) && bounce_acquire(chunk_bytes, bounce, &bounce_cached);
    for (int b = 0; b < 2 && rc == NNLM_OK; b++) {
        if (hipMalloc(&stage[b], chunk_bytes) != hipSuccess) rc = fail(h, NNLM_ERR_HIP, "nnlm_set_matrix: staging buffer (%zu bytes)", chunk_bytes);
        else if (hipHostMalloc(&hp[b], 3 * prep_blocks * sizeof(double)) != hipSuccess || hipEventCreateWithFlags(&ev_done[b], hipEventDisableTiming) != hipSuccess)
            rc = fail(h, NNLM_ERR_HIP, "nnlm_set_matrix: pinned result buffer / event");
    }
    unsigned nthreads = std::thread::hardware_concurrency();
    nthreads = nthreads >= 16 ? 8u : (nthreads >= 4 ? nthreads / 2 : 1u);
    auto host_copy = [&](double *dst, const double *src, size_t bytes) { // a few threads: one saturates ~10 GB/s of memcpy, the DMA takes 4-5x that
        if (nthreads <= 1 || bytes < ((size_t)8 << 20)) {
            memcpy(dst, src, bytes);
            return;
        }
        // (no exception may cross the C ABI: a std::thread that cannot be created -- thread limit of a container -- or a vector that
        //  cannot grow leaves its piece, and every piece after it, to a plain memcpy on this thread, and the mode is kept off from then on)
        std::vector<std::thread> th;
        const size_t per = ((bytes / nthreads) + 4095) & ~(size_t)4095;
        size_t started = 0; // bytes handed to worker threads
        try {
            th.reserve(nthreads);
            for (unsigned t = 0; t < nthreads; t++) {
                const size_t o = (size_t)t * per;
                if (o >= bytes) break;
                const size_t len = (bytes - o < per) ? bytes - o : per;
                th.emplace_back([=]() { memcpy((char *)dst + o, (const char *)src + o, len); });
                started = o + len;
            }
        } catch (...) {
            nthreads = 1;
        }
        if (started < bytes) memcpy((char *)dst + started, (const char *)src + started, bytes - started);
        for (auto &x : th) x.join();
    };
    double cnt = 0.0, klc = 0.0, over = 0.0;
    auto collect = [&](int b) -> int { // wait for slot b's prep pass and add its partial sums (fixed order: chunk by chunk)
        const hipError_t e = hipEventSynchronize(ev_done[b]);
        if (e != hipSuccess) return fail(h, NNLM_ERR_HIP, "prep pass failed: %s", hipGetErrorString(e));
        for (size_t q = 0; q < prep_blocks; q++) {
            cnt += hp[b][3 * q];
            klc += hp[b][3 * q + 1];
            over += hp[b][3 * q + 2];
        }
        return NNLM_OK;
    };
    int pending[2] = {0, 0}, slot = 0;
    for (int j0 = 0; j0 < m && rc == NNLM_OK; j0 += cols_per_chunk, slot ^= 1) {
        const int cols = (m - j0 < cols_per_chunk) ? m - j0 : cols_per_chunk;
        const size_t bytes = (size_t)cols * n * 8;
        if (pending[slot]) { // the slot's previous chunk: its DMA has read the bounce buffer, its prep pass the staging buffer
            rc = collect(slot);
            pending[slot] = 0;
            if (rc != NNLM_OK) break;
        }
        const double *src = A + (size_t)j0 * n;
        if (pinned) {
            host_copy(bounce[slot], src, bytes);
            src = bounce[slot];
        }
        hipError_t e = hipMemcpyAsync(stage[slot], src, bytes, hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) { rc = fail(h, NNLM_ERR_HIP, "upload of A failed: %s", hipGetErrorString(e)); break; }
        dim3 grid(gx, PREP_GRID_Y);
        double *part = h->partials + (size_t)slot * 3 * prep_blocks;
        if (h->prec == NNLM_PREC_F64)
            prep_convert_kernel<double><<<grid, PREP_BLOCK, 0, h->stream>>>(stage[slot], n, cols, j0, (double *)h->A, h->npad, h->miss, part);
        else
            prep_convert_kernel<float><<<grid, PREP_BLOCK, 0, h->stream>>>(stage[slot], n, cols, j0, (float *)h->A, h->npad, h->miss, part);
        e = hipMemcpyAsync(hp[slot], part, 3 * prep_blocks * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipEventRecord(ev_done[slot], h->stream);
        if (e != hipSuccess) { rc = fail(h, NNLM_ERR_HIP, "prep pass failed: %s", hipGetErrorString(e)); break; }
        pending[slot] = 1;
    }
    // (in chunk order: the slot used last holds the newest chunk)
    for (int q = 0; q < 2; q++) {
        const int b = slot ^ q; // slot was flipped past the last chunk: slot = the older one
        if (pending[b] && rc == NNLM_OK) rc = collect(b);
    }
    hipStreamSynchronize(h->stream);
    if (pinned) bounce_give_back(bounce, bounce_cached);
    for (int b = 0; b < 2; b++) {
        hipFree(stage[b]);
        if (hp[b]) hipHostFree(hp[b]);
        if (ev_done[b]) hipEventDestroy(ev_done[b]);
    }
    if (rc != NNLM_OK) return rc;
    if (over > 0.0) {
        free_matrix(h);
        return fail(h, NNLM_ERR_UNSUPPORTED, "%.0f finite entries of A exceed the fp32 range (|a| > 3.4e38): the fp32-operand mode cannot hold them; "
                                             "use the strict fp64 mode (NNLM_PREC_F64, the default of nnlm_c_nnmf / nnlm_c_nnlm)", over);
    }
    h->n_non_missing = cnt;
    h->any_missing = cnt != (double)n * (double)m;
    h->kl_const = klc / cnt; // mean((A+eps) log(A+eps) - A) over finite entries, src/nnmf.cpp:70,73
    if (h->x16) { // split-fp16 copies of A and of its transpose, pre-scaled by 2^eA (k_xprod16.h)
        const size_t cnt_a = (size_t)h->npad * h->mpad;
        HIPCHK(h, hipMemsetAsync(h->maxbits, 0, sizeof(unsigned), h->stream));
        absmax_f32_kernel<<<1024, 256, 0, h->stream>>>((const float *)h->A, cnt_a, h->maxbits);
        unsigned mb = 0;
        HIPCHK(h, hipMemcpyAsync(&mb, h->maxbits, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float mx;
        memcpy(&mx, &mb, 4);
        if (mx > 0.0f && mx < 7.8886090522101181e-31f) { // 2^-100: entries 2^-26 below the largest one are fp32 denormals
            free_matrix(h);
            return fail(h, NNLM_ERR_UNSUPPORTED, "max |A| = %.3g is too close to the bottom of the fp32 range for the fp32-operand mode (entries would "
                                                 "be denormal or flushed to zero); rescale A or use the strict fp64 mode", (double)mx);
        }
        const int eA = split16_exponent(mx);
        HIPCHK(h, hipMemcpyAsync(h->scal_exp, &eA, sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMalloc(&h->A16, cnt_a * 4 + 4096));
        HIPCHK(h, hipMalloc(&h->A16T, cnt_a * 4 + 4096));
        const float scale = ldexpf(1.0f, eA);
        a16_convert_kernel<<<(unsigned)((cnt_a + 255) / 256), 256, 0, h->stream>>>((const float *)h->A, cnt_a, scale, h->A16);
        dim3 gt(h->npad / 64, h->mpad / 64);
        a16_transpose_kernel<<<gt, 256, 0, h->stream>>>((const float *)h->A, h->npad, h->npad, h->mpad, scale, h->A16T);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        LAUNCHCHK(h);
    }
    if (h->any_missing) { // row-wise view of the missing mask for the W half-step
        const size_t wordsT = (size_t)h->npad * (h->mpad / 32);
        HIPCHK(h, hipMalloc(&h->missT, wordsT * 4 + 64));
        dim3 grid((h->mpad / 32 + 255) / 256, h->npad);
        miss_transpose_kernel<<<grid, 256, 0, h->stream>>>(h->miss, h->npad, h->mpad, h->missT);
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return NNLM_OK;
}

extern "C" int nnlm_matrix_info(nnlm_handle *h, double *n_non_missing, int *any_missing, double *kl_const)
{
    if (!h || !h->A) return fail(h, NNLM_ERR_ARG, "nnlm_matrix_info: no matrix set");
    if (n_non_missing) *n_non_missing = h->n_non_missing;
    if (any_missing) *any_missing = h->any_missing ? 1 : 0;
    if (kl_const) *kl_const = h->kl_const;
    return NNLM_OK;
}

// ---------------------------------------------------------------------------------------------
// factors
// ---------------------------------------------------------------------------------------------
// Split-K factor S for tiles_x output tiles.  The cross-product kernels run one block per CU (their LDS ring fills the
// CU), so a launch executes in ceil(blocks/256) rounds of ceil(stages/S) stages each; every block also pays ~3 stage
// times of pipeline fill/drain, and every slab costs fp64 writes here and fp64 reads in the solver that sums the slabs
// (S = 16 instead of 3 at config 2 made the sweep's start-up 0.06 ms longer -- measured).  Pick the cheapest S <= 16.
static int split_plan(int tiles_x, int stages, int *S, int *sps)
{
    const int cus = 256;
    const int smax = 16;
    int best_s = 1;
    double best_cost = 1e300;
    for (int s = 1; s <= smax; s++) {
        if (s > 1 && stages / s < 8) break;
        const long blocks = (long)tiles_x * s;
        const long rounds = (blocks + cus - 1) / cus;
        const int per_block = (stages + s - 1) / s;
        const double cost = (double)rounds * (per_block + 3) + 0.75 * s;
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best_s = s;
        }
    }
    int per = (stages + best_s - 1) / best_s;
    if (per < 1) per = 1;
    *S = (stages + per - 1) / per;
    if (*S < 1) *S = 1;
    *sps = per;
    return 0;
}

struct HalfPlan {
    int stage_begin, stage_end, S, sps, tiles_x;
    int col_off = 0; // first column of the factor being solved that this launch covers (column shards of the multi-GPU NA path)
};

// which = 1: H half-step (contraction over i, A as stored); which = 0: W half-step (contraction over j: the same "TN" kernels on
// the transposed copy of A -- A16T in the split-fp16 mode, AT otherwise -- made once per matrix; the reference transposes A in
// EVERY iteration, src/nnmf.cpp:131)
static HalfPlan plan_half(const nnlm_handle *h, int which, int rank, int nranks)
{
    HalfPlan p;
    int stages_total, tiles_x;
    if (which == 1) {
        const int CE = XPROD_ROWB / (int)esize(h);
        stages_total = h->npad / CE;
        tiles_x = h->mpad / XPROD_TN_BJ;
    } else if (h->x16) { // split-fp16: the TN kernel on the transposed copy, tiles of 128 rows of A, stages of 64 columns
        stages_total = h->mpad / 64;
        tiles_x = h->npad / XPROD_TN_BJ;
    } else { // strict mode (and rank > 64 without split copies): the TN kernel on the transposed copy AT
        stages_total = h->mpad / (XPROD_ROWB / (int)esize(h));
        tiles_x = h->npad / XPROD_TN_BJ;
    }
    // this rank's slab of the contraction
    const int per_rank = (stages_total + nranks - 1) / nranks;
    p.stage_begin = rank * per_rank;
    p.stage_end = p.stage_begin + per_rank;
    if (p.stage_end > stages_total) p.stage_end = stages_total;
    if (p.stage_begin > stages_total) p.stage_begin = stages_total;
    int len = p.stage_end - p.stage_begin;
    if (len < 1) len = 1;
    split_plan(tiles_x, len, &p.S, &p.sps);
    p.tiles_x = tiles_x;
    return p;
}

// contraction elements one stage of the cross product of half-step `which` covers (the granularity of the multi-GPU split)
static int stage_elems(const nnlm_handle *h, int which)
{
    if (which == 1) return XPROD_ROWB / (int)esize(h);
    if (h->x16) return 64;
    return XPROD_ROWB / (int)esize(h);
}

static void pack_mask_cols(const int *mask, int k, int ncols, bool transposed_input, int ld_in, std::vector<unsigned long long> &out, int npadded,
                           int mw)
{
    // transposed_input: mask is ncols x k column-major (Wm, n x k); else k x ncols column-major (Hm); mw words per column
    out.assign((size_t)npadded * mw, 0ull);
    for (int c = 0; c < ncols; c++)
        for (int q = 0; q < k; q++) {
            const int v = transposed_input ? mask[(size_t)q * ld_in + c] : mask[(size_t)c * ld_in + q];
            if (v != 0) out[(size_t)c * mw + (q >> 6)] |= (1ull << (q & 63));
        }
}

extern "C" int nnlm_set_factors(nnlm_handle *h, unsigned k_, const double *W, const double *H, const int *Wm, const int *Hm)
{
    if (!h || !h->A) return fail(h, NNLM_ERR_ARG, "nnlm_set_factors: set the matrix first");
    const int k = (int)k_;
    if (k < 1) return fail(h, NNLM_ERR_ARG, "nnlm_set_factors: rank k must be >= 1");
    HIPCHK(h, hipSetDevice(h->device));
    sync_all(h);
    invalidate_factor_caches(h);
    HIPCHK(h, hipMemset(h->maxbits + 8, 0, sizeof(unsigned)));
    if (k != h->k) {
        free_factors(h);
        h->k = k;
        h->NKQ = (k + 15) / 16; // > 4 beyond rank 64: the generic kernels of k_generic.h take over
        h->KP = 16 * h->NKQ;
        h->KP8 = round_up_i(k, 8);
        h->MW = (k + 63) / 64;
        const size_t es = esize(h);
        for (int i = 0; i < 2; i++) {
            HIPCHK(h, hipMalloc(&h->W64b[i], (size_t)h->KP * h->npad * 8));
            HIPCHK(h, hipMemset(h->W64b[i], 0, (size_t)h->KP * h->npad * 8));
            if (h->prec == NNLM_PREC_F64) h->Wopb[i] = h->W64b[i];
            else {
                HIPCHK(h, hipMalloc(&h->Wopb[i], (size_t)h->KP * h->npad * es));
                HIPCHK(h, hipMemset(h->Wopb[i], 0, (size_t)h->KP * h->npad * es));
            }
        }
        HIPCHK(h, hipMalloc(&h->H64, (size_t)h->KP * h->mpad * 8));
        if (h->prec == NNLM_PREC_F32) HIPCHK(h, hipMalloc(&h->Hkq, (size_t)h->KP * h->mpad * sizeof(float)));
        HIPCHK(h, hipMalloc(&h->Wmask, (size_t)h->npad * h->MW * 8));
        HIPCHK(h, hipMalloc(&h->Hmask, (size_t)h->mpad * h->MW * 8));
        // split-K slabs: sized for the worst case over ranks (nranks = 1 gives the largest S)
        const HalfPlan ph = plan_half(h, 1, 0, 1), pw = plan_half(h, 0, 0, 1);
        size_t eh = (size_t)ph.S * h->KP * h->mpad, ew = (size_t)pw.S * h->KP * h->npad;
        if (h->x16) {
            HIPCHK(h, hipMalloc(&h->Y16, (size_t)h->KP * (h->npad > h->mpad ? h->npad : h->mpad) * 4 + 4096));
            HIPCHK(h, hipMalloc(&h->W16c, (size_t)h->npad * 64 * 4 + 4096));
            HIPCHK(h, hipMalloc(&h->H16c, (size_t)h->mpad * 64 * 4 + 4096));
        }
        h->Cx_elems = eh > ew ? eh : ew;
        HIPCHK(h, hipMalloc(&h->Cx, h->Cx_elems * 8));
        HIPCHK(h, hipMemset(h->Cx, 0, h->Cx_elems * 8)); // rows >= k of a slab are never written: keep them finite
        const int gb = (h->npad > h->mpad ? h->npad : h->mpad) / GRAM_COLS_PER_BLOCK + 1;
        HIPCHK(h, hipMalloc(&h->gslabs, (size_t)gb * h->KP * h->KP * 8));
        HIPCHK(h, hipMalloc(&h->red, ((size_t)h->KP * h->KP + (size_t)h->KP * (h->npad > h->mpad ? h->npad : h->mpad)) * 8));
        h->Graw = h->red;
    }
    h->wcur = 0;
    h->W64 = h->W64b[0];
    h->Wop = h->Wopb[0];
    h->sg_which = h->sg_other = -1;
    h->sw_active = 0;
    HIPCHK(h, hipMemset(h->sweeps, 0, 2 * sizeof(unsigned long long)));
    const int KP = h->KP, n = h->n, m = h->m, npad = h->npad, mpad = h->mpad;
    // host-side repack into the padded resident layouts (k*(n+m) elements: negligible)
    std::vector<double> w64((size_t)KP * npad, 0.0), h64((size_t)KP * mpad, 0.0);
    if (W)
        for (int q = 0; q < k; q++)
            for (int i = 0; i < n; i++) w64[(size_t)q * npad + i] = W[(size_t)q * n + i]; // W is n x k column-major
    if (H)
        for (int j = 0; j < m; j++)
            for (int q = 0; q < k; q++) h64[(size_t)q * mpad + j] = H[(size_t)j * k + q]; // H is k x m column-major
    HIPCHK(h, hipMemcpy(h->W64, w64.data(), w64.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->H64, h64.data(), h64.size() * 8, hipMemcpyHostToDevice));
    if (h->prec != NNLM_PREC_F64) {
        std::vector<float> wop((size_t)KP * npad);
        for (size_t e = 0; e < wop.size(); e++) wop[e] = (float)w64[e];
        HIPCHK(h, hipMemcpy(h->Wop, wop.data(), wop.size() * 4, hipMemcpyHostToDevice));
    }
    std::vector<unsigned long long> mk;
    h->has_wmask = Wm != nullptr;
    h->has_hmask = Hm != nullptr;
    if (Wm) {
        pack_mask_cols(Wm, k, n, true, n, mk, npad, h->MW);
        HIPCHK(h, hipMemcpy(h->Wmask, mk.data(), (size_t)npad * h->MW * 8, hipMemcpyHostToDevice));
    }
    if (Hm) {
        pack_mask_cols(Hm, k, m, false, k, mk, mpad, h->MW);
        HIPCHK(h, hipMemcpy(h->Hmask, mk.data(), (size_t)mpad * h->MW * 8, hipMemcpyHostToDevice));
    }
    return NNLM_OK;
}

extern "C" int nnlm_get_factors(nnlm_handle *h, double *W, double *H)
{
    if (!h || !h->W64) return fail(h, NNLM_ERR_ARG, "nnlm_get_factors: no factors set");
    HIPCHK(h, hipSetDevice(h->device));
    sync_all(h);
    const int k = h->k, n = h->n, m = h->m, npad = h->npad, mpad = h->mpad;
    if (W) {
        std::vector<double> w64((size_t)h->KP * npad);
        HIPCHK(h, hipMemcpy(w64.data(), h->W64, w64.size() * 8, hipMemcpyDeviceToHost));
        for (int q = 0; q < k; q++) memcpy(W + (size_t)q * n, &w64[(size_t)q * npad], (size_t)n * 8);
    }
    if (H) {
        std::vector<double> h64((size_t)h->KP * mpad);
        HIPCHK(h, hipMemcpy(h64.data(), h->H64, h64.size() * 8, hipMemcpyDeviceToHost));
        for (int j = 0; j < m; j++)
            for (int q = 0; q < k; q++) H[(size_t)j * k + q] = h64[(size_t)q * mpad + j];
    }
    return NNLM_OK;
}

// ---------------------------------------------------------------------------------------------
// half-step
// ---------------------------------------------------------------------------------------------
template <typename T, int NKQ, int KT>
static void launch_xprod(nnlm_handle *h, int which, const HalfPlan &p)
{
    const int KP = 16 * (NKQ + (KT > 0 ? 1 : 0)); // = h->KP
    dim3 grid(p.tiles_x, p.S);
    const int lds = xprod_tn_lds_bytes(KP);
    set_dyn_lds((const void *)xprod_tn_kernel<T, NKQ, KT>, lds, "xprod_tn_kernel");
    if (which == 1)
        xprod_tn_kernel<T, NKQ, KT><<<grid, XPROD_THREADS, lds, h->stream>>>((const T *)h->A + (size_t)p.col_off * h->npad, h->npad, (const T *)h->Wop,
                                                                              h->npad, h->Cx + p.col_off, h->mpad, (size_t)KP * h->mpad, p.stage_begin,
                                                                              p.stage_end, p.sps);
    else // roles swapped on the transposed copy (ensure_AT ran): rows of A are the columns solved, H (master, [KP][mpad]) the fixed factor
        xprod_tn_kernel<T, NKQ, KT><<<grid, XPROD_THREADS, lds, h->stream>>>((const T *)h->AT + (size_t)p.col_off * h->mpad, h->mpad, (const T *)h->H64,
                                                                              h->mpad, h->Cx + p.col_off, h->npad, (size_t)KP * h->npad, p.stage_begin,
                                                                              p.stage_end, p.sps);
}

// MFMA tiles / VALU tail rows for rank k: k = 16*NKQ + rem; a remainder of 1..4 rows (with at least one full tile) is
// not padded to a whole 16-wide MFMA tile but handled as 2 or 4 tail rows (k_xprod.h).
template <typename T>
static void launch_xprod_nkq(nnlm_handle *h, int which, const HalfPlan &p)
{
    const int full = h->k / 16, rem = h->k % 16;
    if (full >= 1 && rem >= 1 && rem <= 4) {
        const int kt = rem <= 2 ? 2 : 4;
        switch (full * 10 + kt) {
        case 12: launch_xprod<T, 1, 2>(h, which, p); return;
        case 14: launch_xprod<T, 1, 4>(h, which, p); return;
        case 22: launch_xprod<T, 2, 2>(h, which, p); return;
        case 24: launch_xprod<T, 2, 4>(h, which, p); return;
        case 32: launch_xprod<T, 3, 2>(h, which, p); return;
        default: launch_xprod<T, 3, 4>(h, which, p); return;
        }
    }
    switch (h->NKQ) {
    case 1: launch_xprod<T, 1, 0>(h, which, p); break;
    case 2: launch_xprod<T, 2, 0>(h, which, p); break;
    case 3: launch_xprod<T, 3, 0>(h, which, p); break;
    default: launch_xprod<T, 4, 0>(h, which, p); break;
    }
}

// Split-fp16 cross product (k_xprod16.h): split copy of the fixed factor scaled by its own power of two, then the
// A-streaming kernel on A16 (H half-step) or A16T (W half-step: the transposed copy makes it the same "TN" kernel).
// Y16 / Cx / slab_stride: rows [q0, q0 + 16 NKQ) of the split copy and of the slabs when the rank exceeds 64 (NULL / 0: all of them)
template <int NKQ>
static void launch_xprod16_m(nnlm_handle *h, const uint32_t *A16, int lda, int ldy, int ldc, const HalfPlan &p, const uint32_t *Y16 = nullptr,
                             double *Cx = nullptr, size_t slab_stride = 0)
{
    const int KP = 16 * NKQ;
    dim3 grid(p.tiles_x, p.S);
    const int lds = xprod_tn_lds_bytes(KP);
    set_dyn_lds((const void *)xprod16_tn_kernel<NKQ>, lds, "xprod16_tn_kernel");
    xprod16_tn_kernel<NKQ><<<grid, XPROD_THREADS, lds, h->stream>>>(A16 + (size_t)p.col_off * lda, lda, Y16 ? Y16 : h->Y16, ldy,
                                                                    (Cx ? Cx : h->Cx) + p.col_off, ldc, slab_stride ? slab_stride : (size_t)KP * ldc,
                                                                    p.stage_begin, p.stage_end, p.sps, h->scal_exp);
}
// split copy of the fixed factor, scaled by its own power of two (two small kernels, outside the cross product's timing
// scope).  Measured: making these faster (2-D absmax grid, no memset) or moving sweep_consts_kernel to the Gram stream
// made the step SLOWER by 1 % -- the Gram kernels then overlap more of the (now HBM-bound) cross product.
// mb: device word that already holds max|factor| (from gram_partial_kernel), or NULL: compute it here
// w_max_in_zero_word: zero_word (about to be cleared for this half-step's sweep) still holds max|W| of the current W, left by
// the sweep that solved it -- the fused error block's split copy of W then needs no absmax pass (24 us) of its own.
static void prepare_factor16(nnlm_handle *h, int which, unsigned *mb = nullptr, unsigned *zero_word = nullptr, bool w_max_in_zero_word = false)
{
    const double *Ym = (which == 1) ? h->W64 : h->H64;
    const int ldm = (which == 1) ? h->npad : h->mpad; // leading dimension of the master = padded contraction length
    const int plen_true = (which == 1) ? h->n : h->m;
    h->y16_for = -1; // (Y16 is rewritten)
    if (!mb) {
        mb = h->maxbits;
        hipMemsetAsync(mb, 0, sizeof(unsigned), h->stream);
        absmax_f64_kernel<<<(plen_true + 255) / 256, 256, 0, h->stream>>>(Ym, ldm, plen_true, h->k, mb);
    }
    const size_t cnt = (size_t)h->KP * ldm;
    const bool fuse = which == 0 && h->fuse_err; // the fused error block also needs H and W with kq contiguous (same exponent for H)
    if (fuse && w_max_in_zero_word && zero_word) // (before factor16_kernel clears that word)
        factor16c_kernel<<<h->npad / 64, 256, 0, h->stream>>>(h->W64, h->npad, h->n, h->k, zero_word, h->scal_exp + 2, h->W16c);
    factor16_kernel<<<(unsigned)((cnt + 255) / 256), 256, 0, h->stream>>>(Ym, ldm, plen_true, h->k, h->KP, ldm, mb, h->scal_exp + 1, h->Y16, zero_word);
    if (fuse) {
        factor16c_kernel<<<h->mpad / 64, 256, 0, h->stream>>>(h->H64, h->mpad, h->m, h->k, mb, nullptr, h->H16c);
        if (!(w_max_in_zero_word && zero_word)) {
            // (a word of its own: maxbits[0] may be the fixed factor's maximum that the NA flow's row copy reads later, fixed_maxw)
            unsigned *wmb = h->maxbits + 10;
            hipMemsetAsync(wmb, 0, sizeof(unsigned), h->stream);
            absmax_f64_kernel<<<(h->n + 255) / 256, 256, 0, h->stream>>>(h->W64, h->npad, h->n, h->k, wmb);
            factor16c_kernel<<<h->npad / 64, 256, 0, h->stream>>>(h->W64, h->npad, h->n, h->k, wmb, h->scal_exp + 2, h->W16c);
        }
    }
}
template <int NKQ>
static void launch_xprod16_err_m(nnlm_handle *h, const HalfPlan &p)
{
    dim3 grid(p.tiles_x, p.S);
    const int lds = xprod16_err_lds_bytes(NKQ);
    if (h->any_missing) {
        set_dyn_lds((const void *)xprod16_err_kernel<NKQ, true>, lds, "xprod16_err_kernel");
        xprod16_err_kernel<NKQ, true><<<grid, XPROD_THREADS, lds, h->stream>>>(h->A16T, h->mpad, h->Y16, h->mpad, h->H16c, h->W16c, h->Cx, h->npad,
                                                                               (size_t)16 * NKQ * h->npad, p.stage_begin, p.stage_end, p.sps, h->scal_exp,
                                                                               h->scal_exp + 2, h->n, h->m, h->partials, h->err_zero_word, h->missT,
                                                                               h->mpad / 32, p.col_off);
    } else {
        set_dyn_lds((const void *)xprod16_err_kernel<NKQ, false>, lds, "xprod16_err_kernel");
        xprod16_err_kernel<NKQ, false><<<grid, XPROD_THREADS, lds, h->stream>>>(h->A16T, h->mpad, h->Y16, h->mpad, h->H16c, h->W16c, h->Cx, h->npad,
                                                                                (size_t)16 * NKQ * h->npad, p.stage_begin, p.stage_end, p.sps, h->scal_exp,
                                                                                h->scal_exp + 2, h->n, h->m, h->partials, h->err_zero_word, nullptr, 0,
                                                                                p.col_off);
    }
    h->err_zero_word = nullptr;
    h->fused_nb = p.tiles_x * p.S;
}
static void launch_xprod16(nnlm_handle *h, int which, const HalfPlan &p)
{
    if (which == 0 && h->fuse_err) {
        switch (h->NKQ) {
        case 1: launch_xprod16_err_m<1>(h, p); break;
        case 2: launch_xprod16_err_m<2>(h, p); break;
        case 3: launch_xprod16_err_m<3>(h, p); break;
        default: launch_xprod16_err_m<4>(h, p); break;
        }
        return;
    }
    const int ldm = (which == 1) ? h->npad : h->mpad;
    const uint32_t *A16 = (which == 1) ? h->A16 : h->A16T;
    const int ldc = (which == 1) ? h->mpad : h->npad;
    switch (h->NKQ) {
    case 1: launch_xprod16_m<1>(h, A16, ldm, ldm, ldc, p); break;
    case 2: launch_xprod16_m<2>(h, A16, ldm, ldm, ldc, p); break;
    case 3: launch_xprod16_m<3>(h, A16, ldm, ldm, ldc, p); break;
    default: launch_xprod16_m<4>(h, A16, ldm, ldm, ldc, p); break;
    }
}

// ---- rank > 64 (k_generic.h): the same A-streaming kernels, launched once per 64 rows of the fixed factor --------------
static bool generic_rank(const nnlm_handle *h) { return h->k > NNLM_KQ_MAX; }

// contraction-contiguous copy of A ([npad][mpad], element type of the mode), made once per matrix
// soft: a copy that does not fit is reported as NNLM_ERR_UNSUPPORTED without an error message (the KL solvers then take their streaming path)
static int ensure_AT(nnlm_handle *h, bool soft = false)
{
    if (h->AT) return NNLM_OK;
    const size_t bytes = (size_t)h->npad * h->mpad * esize(h) + 4096;
    if (soft) {
        if (big_malloc(&h->AT, bytes) != hipSuccess) return NNLM_ERR_UNSUPPORTED;
    } else
        HIPCHK(h, hipMalloc(&h->AT, bytes));
    dim3 grid(h->npad / 64, h->mpad / 64);
    if (h->prec == NNLM_PREC_F64) transpose_kernel<double><<<grid, 256, 0, h->stream>>>((const double *)h->A, h->npad, (double *)h->AT, h->mpad);
    else transpose_kernel<float><<<grid, 256, 0, h->stream>>>((const float *)h->A, h->npad, (float *)h->AT, h->mpad);
    LAUNCHCHK(h);
    return NNLM_OK;
}

template <typename T, int NKQ>
static void launch_xprod_tn_rows(nnlm_handle *h, const T *Amat, int lda, const T *Y, int ldy, double *C, int ldc, size_t slab_stride, const HalfPlan &p)
{
    dim3 grid(p.tiles_x, p.S);
    const int lds = xprod_tn_lds_bytes(16 * NKQ);
    set_dyn_lds((const void *)xprod_tn_kernel<T, NKQ, 0>, lds, "xprod_tn_kernel");
    xprod_tn_kernel<T, NKQ, 0><<<grid, XPROD_THREADS, lds, h->stream>>>(Amat + (size_t)p.col_off * lda, lda, Y, ldy, C + p.col_off, ldc, slab_stride,
                                                                         p.stage_begin, p.stage_end, p.sps);
}

template <typename T>
static int launch_xprod_generic_t(nnlm_handle *h, int which, const HalfPlan &p)
{
    const int KP = h->KP;
    const T *Amat;
    const T *Y;
    int lda, ldy, ldc;
    if (which == 1) {
        Amat = (const T *)h->A; lda = h->npad; Y = (const T *)h->Wop; ldy = h->npad; ldc = h->mpad;
    } else {
        int rc = ensure_AT(h);
        if (rc != NNLM_OK) return rc;
        Amat = (const T *)h->AT; lda = h->mpad; ldy = h->mpad; ldc = h->npad;
        if (h->prec == NNLM_PREC_F64) Y = (const T *)h->H64;
        else { // fp32 [KP][mpad] copy of H
            const size_t cnt = (size_t)KP * h->mpad;
            factor_to_f32_kernel<<<(unsigned)((cnt + 255) / 256), 256, 0, h->stream>>>(h->H64, cnt, h->Hkq);
            Y = (const T *)h->Hkq;
        }
    }
    const size_t slab = (size_t)KP * ldc;
    for (int q0 = 0; q0 < KP; q0 += 64) {
        const int nk = (KP - q0 >= 64) ? 4 : (KP - q0) / 16;
        const T *Yq = Y + (size_t)q0 * ldy;
        double *Cq = h->Cx + (size_t)q0 * ldc;
        switch (nk) {
        case 1: launch_xprod_tn_rows<T, 1>(h, Amat, lda, Yq, ldy, Cq, ldc, slab, p); break;
        case 2: launch_xprod_tn_rows<T, 2>(h, Amat, lda, Yq, ldy, Cq, ldc, slab, p); break;
        case 3: launch_xprod_tn_rows<T, 3>(h, Amat, lda, Yq, ldy, Cq, ldc, slab, p); break;
        default: launch_xprod_tn_rows<T, 4>(h, Amat, lda, Yq, ldy, Cq, ldc, slab, p); break;
        }
    }
    return NNLM_OK;
}

static int launch_xprod_generic(nnlm_handle *h, int which, const HalfPlan &p)
{
    if (h->x16) { // split-fp16 copies A16 / A16T exist: the split copy of the factor (prepare_factor16) covers all KP rows
        const int KP = h->KP;
        const int ldm = (which == 1) ? h->npad : h->mpad, ldc = (which == 1) ? h->mpad : h->npad;
        const uint32_t *A16 = (which == 1) ? h->A16 : h->A16T;
        const size_t slab = (size_t)KP * ldc;
        for (int q0 = 0; q0 < KP; q0 += 64) {
            const int nk = (KP - q0 >= 64) ? 4 : (KP - q0) / 16;
            const uint32_t *Yq = h->Y16 + (size_t)q0 * ldm;
            double *Cq = h->Cx + (size_t)q0 * ldc;
            switch (nk) {
            case 1: launch_xprod16_m<1>(h, A16, ldm, ldm, ldc, p, Yq, Cq, slab); break;
            case 2: launch_xprod16_m<2>(h, A16, ldm, ldm, ldc, p, Yq, Cq, slab); break;
            case 3: launch_xprod16_m<3>(h, A16, ldm, ldm, ldc, p, Yq, Cq, slab); break;
            default: launch_xprod16_m<4>(h, A16, ldm, ldm, ldc, p, Yq, Cq, slab); break;
            }
        }
        return NNLM_OK;
    }
    if (h->prec == NNLM_PREC_F64) return launch_xprod_generic_t<double>(h, which, p);
    return launch_xprod_generic_t<float>(h, which, p);
}

// mb (optional): zeroed word that receives the bit pattern of max|Y| over the range (the split-fp16 copy's scale: no absmax pass)
static void launch_gram(nnlm_handle *h, const double *Y, int ld, int c_begin, int c_end, int *nslabs, unsigned *mb = nullptr)
{
    int nb = (c_end - c_begin + GRAM_COLS_PER_BLOCK - 1) / GRAM_COLS_PER_BLOCK;
    if (nb < 1) nb = 1;
    if (generic_rank(h)) {
        dim3 grid(nb, h->NKQ * (h->NKQ + 1) / 2);
        gram_partial_generic_kernel<<<grid, 256, 0, h->stream>>>(Y, ld, c_begin, c_end, h->NKQ, h->gslabs);
        gram_reduce_kernel<<<(h->KP * h->KP + 255) / 256, 256, 0, h->stream>>>(h->gslabs, nb, h->KP, h->Graw);
        *nslabs = nb;
        return;
    }
    switch (h->NKQ) {
    case 1: gram_partial_kernel<1><<<nb, 256, 0, h->stream>>>(Y, ld, c_begin, c_end, h->gslabs, mb); break;
    case 2: gram_partial_kernel<2><<<nb, 256, 0, h->stream>>>(Y, ld, c_begin, c_end, h->gslabs, mb); break;
    case 3: gram_partial_kernel<3><<<nb, 256, 0, h->stream>>>(Y, ld, c_begin, c_end, h->gslabs, mb); break;
    default: gram_partial_kernel<4><<<nb, 256, 0, h->stream>>>(Y, ld, c_begin, c_end, h->gslabs, mb); break;
    }
    const int KP = h->KP;
    // (16 wavefronts per 64 entries, every 16th slab each: 5 us where the one-thread-per-entry sum over ~80 slabs took 20-25)
    gram_fold_kernel<<<KP * KP / 64, 1024, 0, h->stream>>>(h->gslabs, nb, KP, h->Graw, SweepImg{});
    *nslabs = nb;
}

template <int R, int L>
static void launch_sweep_m(int method, const SweepArgs &a, hipStream_t s)
{
    const int cpw = 64 / L; // columns per wavefront
    const int nb = (a.ncols + cpw - 1) / cpw;
    (void)method; // Lee's multiplicative updates only: SCD-LS is sweep_scd_q_kernel at every rank <= 64 (launch_sweep)
    sweep_ls_kernel<R, L, 2><<<nb, 64, 0, s>>>(a);
}

// registers per lane R = STEP * idx, idx = 1..8
template <int STEP, int L>
static void launch_sweep_l(int idx, int method, const SweepArgs &a, hipStream_t s)
{
    switch (idx) {
    case 1: launch_sweep_m<1 * STEP, L>(method, a, s); break;
    case 2: launch_sweep_m<2 * STEP, L>(method, a, s); break;
    case 3: launch_sweep_m<3 * STEP, L>(method, a, s); break;
    case 4: launch_sweep_m<4 * STEP, L>(method, a, s); break;
    case 5: launch_sweep_m<5 * STEP, L>(method, a, s); break;
    case 6: launch_sweep_m<6 * STEP, L>(method, a, s); break;
    case 7: launch_sweep_m<7 * STEP, L>(method, a, s); break;
    default: launch_sweep_m<8 * STEP, L>(method, a, s); break;
    }
}

// Lanes per column.  The sweep is bound by the per-wavefront issue rate (fp64 VALU: one instruction per 8 cycles per
// wave, measured), so fewer FMAs per lane (larger L) wins as long as the SIMDs are not oversubscribed: L = 4 up to
// 4 wavefronts per SIMD (measured at config 2: L=4 0.56 ms, L=2 0.85 ms, L=1 1.5 ms per half-step).
static int sweep_lanes_per_column(int ncols)
{
    const long slots = 1024L * 4; // SIMDs x wavefronts per SIMD
    if (((long)ncols * 4 + 63) / 64 <= slots) return 4;
    if (((long)ncols * 2 + 63) / 64 <= slots) return 2;
    return 1;
}

// SCD-LS for ranks up to 64, both modes: sweep_scd_q_kernel (k_sweep_q.h: one wavefront per 16 columns, the whole recurrence on the
// 4x4x4 fp64 matrix instruction).  fp32-operand mode: rows of G divided by their diagonal, one instruction per chain pass; strict
// fp64 mode: the reference's arithmetic (correctly rounded mu / G[q][q]).
static bool sweep_fast(const nnlm_handle *h) { return h->prec == NNLM_PREC_F32 && h->k <= NNLM_KQ_MAX; }

// (the instantiations of sweep_scd_q_kernel / sweep_scd_qw_kernel / sweep_scd_f_kernel -- 64 heavy ones each -- live in translation units of
//  their own, tu_sweepq.hip / tu_sweepqw.hip / tu_sweepf.hip, compiled next to this one: tu_sweepq.h)
// fp32-operand mode: the fp32-chain kernel (k_sweep_f.h).  One wavefront per 16 columns; four per workgroup (one per SIMD) while the
// launch has at most one wavefront per SIMD of the device, eight (two per SIMD, 128 columns per workgroup) beyond -- the instruction
// streams of two wavefronts interleave where a lone one leaves every second issue slot empty.  Reads a.Graw; no operand image.
// Gram partial-sum slabs a sweep launch may leave behind: one per workgroup -- 64 columns (k_sweep_q.h), 64 / 128 (k_sweep_f.h), or 16 / 32 in
// the row form, which is taken for at most one workgroup per CU
static size_t sg_slab_count(const nnlm_handle *h)
{
    const int big = h->n > h->m ? h->n : h->m;
    const int wide = (big + SWEEPQ_COLS - 1) / SWEEPQ_COLS, row = (big + 15) / 16 < h->cus ? (big + 15) / 16 : h->cus;
    return (size_t)(wide > row ? wide : row) + 1;
}
static void launch_sweep_f(nnlm_handle *h, const SweepArgs &a)
{
    const int ncols = a.ncols - a.col0, NB = (a.k + 3) / 4;
    h->sweep_wgs = 0;
    h->pack_ready = false;
    if (ncols <= 0) return;
    // Row form (k_sweep_r.h: four columns per wavefront, no matrix instruction in the chain) while the launch is ONE round of its
    // wavefronts: workgroups of four wavefronts (one per SIMD) up to one workgroup per CU, of eight (two per SIMD) up to 32 columns per CU.
    // Mid-size problems and the column shards of a multi-GPU run; beyond (the benchmark's 10000 / 20000 columns on one GPU) the
    // matrix-pipe kernel below, whose 157 workgroups are one round where the row form's would be two or three.
    if (a.k <= SWEEPR_KMAX && ncols <= 32 * h->cus) {
        const int NWr = (ncols <= 16 * h->cus) ? 4 : 8;
        const int nbr = (ncols + 4 * NWr - 1) / (4 * NWr);
        h->sweep_wgs = nbr;
        h->sweep_form[h->cur_which] = 3;
        h->sweep_groups[h->cur_which] = NWr;
        nnlm_tu_sweep_r(a, nbr, NWr, h->stream);
        return;
    }
    const int ngroups = (ncols + 15) / 16, simds = 4 * h->cus;
    const int NW = ngroups > simds ? 8 : 4;
    const int nb = (ncols + 16 * NW - 1) / (16 * NW);
    h->sweep_wgs = nb;
    h->sweep_form[h->cur_which] = 2;
    h->sweep_groups[h->cur_which] = NW;
    const hipError_t ea = nnlm_tu_sweep_f(a, nb, NB, NW, h->stream);
    if (ea != hipSuccess && g_attr_err == hipSuccess) {
        g_attr_err = ea;
        g_attr_what = "sweep_scd_f_kernel";
    }
}

static void launch_sweep_q(nnlm_handle *h, const SweepArgs &a)
{
    if (h->prec == NNLM_PREC_F32) return launch_sweep_f(h, a);
    int nb = (a.ncols - a.col0 + SWEEPQ_COLS - 1) / SWEEPQ_COLS;
    const int NB = (a.k + 3) / 4;
    h->sweep_wgs = 0;
    if (nb <= 0) return;
    if (!h->pack_ready)
        sweepq_pack_kernel<<<8, 256, 0, h->stream>>>(a.Graw, a.KPg, a.k, a.r0, a.r1, NB, h->sweepq_img, 1);
    h->pack_ready = false;
    // Launch policy.  A SIMD runs one wavefront (16 columns) of this sweep at full speed and a second one adds its whole time: the plain
    // form costs ceil(groups / SIMDs) rounds of S sweeps.  The persistent form gives every CU G column groups, shared by its four
    // wavefronts by the wrap-around rule -- ceil(G S / 4) sweeps per round of (one workgroup per CU) -- and is taken, with the G that
    // costs least, whenever that is less: the benchmark's W half-step (1250 groups on 1024 SIMDs) G = 5, 1.25 S instead of 2 S;
    // 2500 groups G = 10, 2.5 S instead of 3 S.  (Costs in quarters of S.)
    const int ngroups = (a.ncols - a.col0 + 15) / 16, simds = 4 * h->cus;
    int G = 0;
    if (ngroups > simds && a.max_iter >= 4) {
        long best = ((long)ngroups + simds - 1) / simds * 4; // the plain form
        for (int g = 5; g <= SWEEPQ_WRAP_MAXG; g++) {
            if (sweepqw_lds_bytes(h->KP, NB, true, g) > (size_t)160 * 1024) break;
            const long wgs = ((long)ngroups + g - 1) / g, rounds = (wgs + h->cus - 1) / h->cus;
            if (rounds * g < best) best = rounds * g, G = g;
        }
    }
    if (G) nb = (ngroups + G - 1) / G;
    h->sweep_wgs = nb;
    h->sweep_form[h->cur_which] = G ? 1 : 0;
    h->sweep_groups[h->cur_which] = G ? G : 4;
    const hipError_t ea = G ? nnlm_tu_sweep_qw(a, h->sweepq_img, nb, NB, true, G, h->stream) : nnlm_tu_sweep_q(a, h->sweepq_img, nb, NB, true, h->stream);
    if (ea != hipSuccess && g_attr_err == hipSuccess) {
        g_attr_err = ea;
        g_attr_what = G ? "sweep_scd_qw_kernel" : "sweep_scd_q_kernel";
    }
}

// rank > 64: one wavefront per column, coordinates in LDS (k_generic.h); g_stride != 0: per-column Grams (missing values)
static int launch_sweep_generic(nnlm_handle *h, int method, const SweepArgs &a, size_t g_stride)
{
    const int ncols = a.ncols - a.col0;
    if (ncols <= 0) return NNLM_OK;
    const bool g_in_lds = g_stride == 0 && sweep_generic_lds_bytes(a.k, a.KPg, true) <= (size_t)160 * 1024;
    const size_t lds = sweep_generic_lds_bytes(a.k, a.KPg, g_in_lds);
    if (lds > (size_t)160 * 1024) return fail(h, NNLM_ERR_UNSUPPORTED, "rank %d needs %zu bytes of LDS per workgroup (limit 160 KiB)", a.k, lds);
    const int nb = (ncols + 3) / 4;
    if (method == 1) {
        set_dyn_lds((const void *)sweep_generic_kernel<1>, (int)lds, "sweep_generic_kernel");
        sweep_generic_kernel<1><<<nb, 256, lds, h->stream>>>(a, g_stride, h->MW, g_in_lds ? 1 : 0);
    } else {
        set_dyn_lds((const void *)sweep_generic_kernel<2>, (int)lds, "sweep_generic_kernel");
        sweep_generic_kernel<2><<<nb, 256, lds, h->stream>>>(a, g_stride, h->MW, g_in_lds ? 1 : 0);
    }
    return NNLM_OK;
}

static int launch_sweep(nnlm_handle *h, int method, const SweepArgs &a)
{
    if (generic_rank(h)) return launch_sweep_generic(h, method, a, 0);
    if (method == 1) {
        launch_sweep_q(h, a);
        return NNLM_OK;
    }
    // Lee's multiplicative updates: sweep_ls_kernel, L lanes per column
    const int L = sweep_lanes_per_column(a.ncols);
    const int rneed = (h->k + L - 1) / L;
    if (L == 4) launch_sweep_l<2, 4>((rneed + 1) / 2, method, a, h->stream);      // R = 2..16, k <= 64
    else if (L == 2) launch_sweep_l<4, 2>((rneed + 3) / 4, method, a, h->stream); // R = 4..32
    else launch_sweep_l<8, 1>((rneed + 7) / 8, method, a, h->stream);             // R = 8..64
    return NNLM_OK;
}

// strict fp64 mode: kl_reg64_kernel (k_kl.h).  EPT2 = double2 chunks of the contraction per thread (rounded up to an
// instantiated value), C = columns per workgroup (C * EPT2 <= 20 chunks = 160 state registers)
template <int EPT2, int C>
static void launch_kl64_m(int method, const Kl64Args &ka, hipStream_t s)
{
    const int nb = (ka.ncols - ka.colbase + C - 1) / C;
    if (nb <= 0) return;
    const size_t lds = kl64_lds_bytes(EPT2, C, ka.k);
    if (method == 3) {
        set_dyn_lds((const void *)kl_reg64_kernel<EPT2, C, 3>, (int)lds, "kl_reg64_kernel");
        kl_reg64_kernel<EPT2, C, 3><<<nb, KL64_THREADS, lds, s>>>(ka);
    } else {
        set_dyn_lds((const void *)kl_reg64_kernel<EPT2, C, 4>, (int)lds, "kl_reg64_kernel");
        kl_reg64_kernel<EPT2, C, 4><<<nb, KL64_THREADS, lds, s>>>(ka);
    }
}
static int kl64_ept2(int p) { return ((p + 1) / 2 + KL64_THREADS - 1) / KL64_THREADS; } // > 20: too long for the register-resident kernel
static bool kl64_fits(int p, int k) { return kl64_ept2(p) <= 20 && kl64_lds_bytes(kl64_ept2(p) <= 5 ? 5 : (kl64_ept2(p) <= 10 ? 10 : 20), 4, k) <= (size_t)160 * 1024; }
static void launch_kl64(int method, const Kl64Args &ka, hipStream_t s)
{
    const int e = kl64_ept2(ka.p);
    if (e <= 1) launch_kl64_m<1, 4>(method, ka, s);
    else if (e <= 2) launch_kl64_m<2, 4>(method, ka, s);
    else if (e <= 3) launch_kl64_m<3, 4>(method, ka, s);
    else if (e <= 5) launch_kl64_m<5, 4>(method, ka, s);
    else if (e <= 7) launch_kl64_m<7, 2>(method, ka, s);
    else if (e <= 10) launch_kl64_m<10, 2>(method, ka, s);
    else if (e <= 12) launch_kl64_m<12, 1>(method, ka, s);
    else if (e <= 14) launch_kl64_m<14, 1>(method, ka, s);
    else if (e <= 16) launch_kl64_m<16, 1>(method, ka, s);
    else if (e <= 18) launch_kl64_m<18, 1>(method, ka, s);
    else launch_kl64_m<20, 1>(method, ka, s);
}

// F32 mode: kl_tile_kernel (k_kl.h).  EPT4 = float4 chunks of the contraction per thread, C = columns per workgroup
// (C * EPT4 * 8 state registers per thread).
template <int EPT4, int C, bool ONEBUF = false>
static void launch_kl_tile_m(int method, const KlTileArgs &ta, int nb, size_t lds, hipStream_t s)
{
    if (method == 3) {
        set_dyn_lds((const void *)kl_tile_kernel<EPT4, C, 3, ONEBUF>, (int)lds, "kl_tile_kernel");
        kl_tile_kernel<EPT4, C, 3, ONEBUF><<<nb, KLT_THREADS, lds, s>>>(ta);
    } else {
        set_dyn_lds((const void *)kl_tile_kernel<EPT4, C, 4, ONEBUF>, (int)lds, "kl_tile_kernel");
        kl_tile_kernel<EPT4, C, 4, ONEBUF><<<nb, KLT_THREADS, lds, s>>>(ta);
    }
}
// up to 10 pieces per thread: two row buffers, 2 .. 8 columns per block; 11 .. 20 (contractions up to ~40400): one row buffer, one column
static int kl_tile_cols(int ept4) { return ept4 <= 2 ? 8 : (ept4 <= 5 ? 4 : (ept4 <= 10 ? 2 : 1)); }
static int kl_tile_nbuf(int ept4) { return ept4 <= 10 ? 2 : 1; }
static int kl_tile_ept4(int p)
{
    const int e = (kl_tile_p4(p) + KLT_THREADS - 1) / KLT_THREADS; // float4 pieces per thread: the instantiation is exact
    return e <= 20 ? e : 0;                                        // 0: contraction too long for the register-resident kernel
}
static bool kl_tile_fits(int p, int k, int mw_masked)
{
    const int e = kl_tile_ept4(p);
    return e > 0 && kl_tile_lds_bytes(p, k, kl_tile_cols(e), mw_masked, kl_tile_nbuf(e)) <= (size_t)160 * 1024 &&
           2 * round_up_i(k, 2) * ERRF_TILE * 4 <= 160 * 1024;
}
static void launch_kl_tile(int method, const KlTileArgs &ta, hipStream_t s)
{
    const int e = kl_tile_ept4(ta.p), C = kl_tile_cols(e);
    const int nb = (ta.ncols - ta.colbase + C - 1) / C;
    if (nb <= 0) return;
    const size_t lds = kl_tile_lds_bytes(ta.p, ta.k, C, ta.mask ? ta.mw : 0, kl_tile_nbuf(e));
    switch (e) {
    case 11: launch_kl_tile_m<11, 1, true>(method, ta, nb, lds, s); break;
    case 12: launch_kl_tile_m<12, 1, true>(method, ta, nb, lds, s); break;
    case 13: launch_kl_tile_m<13, 1, true>(method, ta, nb, lds, s); break;
    case 14: launch_kl_tile_m<14, 1, true>(method, ta, nb, lds, s); break;
    case 15: launch_kl_tile_m<15, 1, true>(method, ta, nb, lds, s); break;
    case 16: launch_kl_tile_m<16, 1, true>(method, ta, nb, lds, s); break;
    case 17: launch_kl_tile_m<17, 1, true>(method, ta, nb, lds, s); break;
    case 18: launch_kl_tile_m<18, 1, true>(method, ta, nb, lds, s); break;
    case 19: launch_kl_tile_m<19, 1, true>(method, ta, nb, lds, s); break;
    case 20: launch_kl_tile_m<20, 1, true>(method, ta, nb, lds, s); break;
    case 1: launch_kl_tile_m<1, 8>(method, ta, nb, lds, s); break;
    case 2: launch_kl_tile_m<2, 8>(method, ta, nb, lds, s); break;
    case 3: launch_kl_tile_m<3, 4>(method, ta, nb, lds, s); break;
    case 4: launch_kl_tile_m<4, 4>(method, ta, nb, lds, s); break;
    case 5: launch_kl_tile_m<5, 4>(method, ta, nb, lds, s); break;
    case 6: launch_kl_tile_m<6, 2>(method, ta, nb, lds, s); break;
    case 7: launch_kl_tile_m<7, 2>(method, ta, nb, lds, s); break;
    case 8: launch_kl_tile_m<8, 2>(method, ta, nb, lds, s); break;
    case 9: launch_kl_tile_m<9, 2>(method, ta, nb, lds, s); break;
    default: launch_kl_tile_m<10, 2>(method, ta, nb, lds, s); break;
    }
}

template <typename T>
static int launch_kl_stream(nnlm_handle *h, int method, const KlArgs &a, int mw, void *st, size_t ldst, hipStream_t s)
{
    const size_t lds = (size_t)(a.k + 24) * 8;
    if (a.ncols <= a.col0) return NNLM_OK;
    if (lds > (size_t)160 * 1024)
        return fail(h, NNLM_ERR_UNSUPPORTED, "KL solver: rank %d needs %zu bytes of LDS per workgroup (limit 160 KiB)", a.k, lds);
    if (method == 3) {
        HIPCHK(h, hipFuncSetAttribute((const void *)kl_stream_kernel<T, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kl_stream_kernel<T, 3><<<a.ncols - a.col0, 256, lds, s>>>(a, mw, (T *)st, ldst);
    } else {
        HIPCHK(h, hipFuncSetAttribute((const void *)kl_stream_kernel<T, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kl_stream_kernel<T, 4><<<a.ncols - a.col0, 256, lds, s>>>(a, mw, (T *)st, ldst);
    }
    return NNLM_OK;
}

template <int NKQ>
static void launch_colsolve_m(const SweepArgs &a, size_t g_stride, hipStream_t s) // Lee's multiplicative updates, per-column Grams
{
    const int nb = (a.ncols - a.col0 + 3) / 4;
    if (nb <= 0) return;
    colsolve_ls_kernel<NKQ, 2><<<nb, 256, 0, s>>>(a, g_stride);
}

// F32 mode, SCD: colsolve_row_kernel (k_colsolve_row.h, tu_colsolve.hip): four columns per wavefront, the step's delta by DPP row broadcast
static bool colsolve_fast_ok(const nnlm_handle *h, int method)
{
    return method == 1 && h->prec == NNLM_PREC_F32 && h->k <= NNLM_KQ_MAX;
}

template <int NKQ>
static void launch_colsolve_strict_m(const SweepArgs &a, size_t g_stride, hipStream_t s)
{
    const int nb = (a.ncols - a.col0 + 3) / 4;
    if (nb <= 0) return;
    constexpr int KS = NKQ > 1 ? 16 * (NKQ - 1) + 4 : 16;
    if (NKQ > 1 && a.k <= KS) {
        if (a.mask) colsolve_strict_kernel<NKQ, true, KS><<<nb, 256, 0, s>>>(a, g_stride);
        else colsolve_strict_kernel<NKQ, false, KS><<<nb, 256, 0, s>>>(a, g_stride);
        return;
    }
    if (a.mask) colsolve_strict_kernel<NKQ, true><<<nb, 256, 0, s>>>(a, g_stride);
    else colsolve_strict_kernel<NKQ, false><<<nb, 256, 0, s>>>(a, g_stride);
}
static void launch_colsolve(nnlm_handle *h, int method, const SweepArgs &a, size_t g_stride)
{
    if (colsolve_fast_ok(h, method)) {
        nnlm_tu_colsolve_row(a, g_stride, h->stream);
        return;
    }
    // SCD in the reference's arithmetic (strict mode): the unrolled lane-local form; Lee's updates: colsolve_lee_kernel
    if (method == 1) {
        switch (h->NKQ) {
        case 1: launch_colsolve_strict_m<1>(a, g_stride, h->stream); break;
        case 2: launch_colsolve_strict_m<2>(a, g_stride, h->stream); break;
        case 3: launch_colsolve_strict_m<3>(a, g_stride, h->stream); break;
        default: launch_colsolve_strict_m<4>(a, g_stride, h->stream); break;
        }
        return;
    }
    switch (h->NKQ) {
    case 1: launch_colsolve_m<1>(a, g_stride, h->stream); break;
    case 2: launch_colsolve_m<2>(a, g_stride, h->stream); break;
    case 3: launch_colsolve_m<3>(a, g_stride, h->stream); break;
    default: launch_colsolve_m<4>(a, g_stride, h->stream); break;
    }
}

// CSR row lists of orientation `which` (see nnlm_handle::na_ptr): count, prefix sum on the host (once), fill
static int ensure_na_lists(nnlm_handle *h, int which, const uint32_t *bits, int words, int p, int ncols)
{
    if (h->na_ptr[which]) return NNLM_OK;
    uint32_t *cnt = nullptr;
    HIPCHK(h, hipMalloc(&cnt, (size_t)ncols * 4));
    na_count_kernel<<<(ncols + 3) / 4, 256, 0, h->stream>>>(bits, words, p, ncols, cnt);
    std::vector<uint32_t> hc(ncols), hptr(ncols + 1), hmeta(ncols);
    HIPCHK(h, hipMemcpyAsync(hc.data(), cnt, (size_t)ncols * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    hipFree(cnt);
    size_t total = 0;
    for (int c = 0; c < ncols; c++) {
        const uint32_t miss = hc[c];
        const bool complement = (size_t)miss * 2 <= (size_t)p; // sum over the missing rows and subtract from the full Gram
        const uint32_t len = complement ? miss : (uint32_t)p - miss;
        hptr[c] = (uint32_t)total;
        hmeta[c] = len | (complement ? 0x80000000u : 0u);
        total += len;
    }
    hptr[ncols] = (uint32_t)total;
    if (total >= 0x7FFFFFFFull) return fail(h, NNLM_ERR_UNSUPPORTED, "missing-value row lists exceed 2^31 entries");
    HIPCHK(h, hipMalloc(&h->na_ptr[which], (size_t)(ncols + 1) * 4));
    HIPCHK(h, hipMalloc(&h->na_meta[which], (size_t)ncols * 4));
    HIPCHK(h, hipMalloc(&h->na_idx[which], (total + 128) * 4)); // (slack: the Gram kernels request whole groups / steps of indices ahead of the end)
    HIPCHK(h, hipMemcpyAsync(h->na_ptr[which], hptr.data(), (size_t)(ncols + 1) * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->na_meta[which], hmeta.data(), (size_t)ncols * 4, hipMemcpyHostToDevice, h->stream));
    na_fill_kernel<<<ncols, 256, 0, h->stream>>>(bits, words, p, h->na_ptr[which], h->na_meta[which], h->na_idx[which]);
    HIPCHK(h, hipStreamSynchronize(h->stream)); // hptr / hmeta go out of scope
    return NNLM_OK;
}

// Row-major copy of the fixed factor the NA / KL-with-NA paths gather from: [p][KP] doubles or floats, or -- split-fp16 Grams --
// [p + 64][64 hi | 64 lo halves] (256 bytes per row whatever KP is; the rows behind p are zero)
static size_t yrow_bytes(const nnlm_handle *h)
{
    const size_t big = (size_t)(h->npad > h->mpad ? h->npad : h->mpad);
    const size_t a = big * h->KP * 8, b = (big + 64) * 256;
    return a > b ? a : b;
}

// Per-column Grams of columns [c0, c1) (row lists exist for all ncols columns)
static int launch_na_gram(nnlm_handle *h, int which, const uint32_t *bits, int words, int p, int ncols, int c0, int c1, bool upper_only = false)
{
    const int nc = c1 - c0;
    if (nc <= 0) return NNLM_OK;
    const double *Ym = (which == 1) ? h->W64 : h->H64; // fixed factor [KP][ldy]; the kernels gather its ROWS: row-major copy Yrow [p][KP]
    const int ldy = (which == 1) ? h->npad : h->mpad;
    int rc = ensure_na_lists(h, which, bits, words, p, ncols);
    if (rc != NNLM_OK) return rc;
    // F32 mode: the Grams on the fp16 matrix cores from the split copy of the rows (na_gram_f16_kernel).
    // max|fixed factor| is in *fixed_maxw (prepare_factor16 ran for this half-step's cross product).
    if (h->x16 && !generic_rank(h)) {
        // [p + 64 rows][64 hi | 64 lo halves]; rows p .. are zero (the kernel's "no row" index)
        factor16c_kernel<<<p / 64 + 1, 256, 0, h->stream>>>(Ym, ldy, p, h->k, h->fixed_maxw ? h->fixed_maxw : h->maxbits, h->scal_exp + 3, (uint32_t *)h->Yrow);
        const int nb = (nc + 3) / 4;
#define NNLM_NAGH(N_) na_gram_f16_kernel<N_><<<nb, 256, 0, h->stream>>>(h->na_ptr[which], h->na_meta[which], h->na_idx[which], (const uint32_t *)h->Yrow, p, h->scal_exp + 3, h->Graw, h->Gcols, c1, c0, h->k, upper_only ? 1 : 0)
        switch (h->NKQ) {
        case 1: NNLM_NAGH(1); break;
        case 2: NNLM_NAGH(2); break;
        case 3: NNLM_NAGH(3); break;
        default: NNLM_NAGH(4); break;
        }
#undef NNLM_NAGH
        return NNLM_OK;
    }
    factor_rows_kernel<double><<<(p + 255) / 256, 256, 0, h->stream>>>(Ym, ldy, p, h->KP, h->Yrow);
    if (generic_rank(h)) { // rank > 64: k_generic.h
        const int lds = 16 * h->KP * 8;
        na_gram_generic_kernel<<<nc, 256, lds, h->stream>>>(h->na_ptr[which], h->na_meta[which], h->na_idx[which], h->Yrow, h->KP, h->Graw, h->Gcols, c0);
        return NNLM_OK;
    }
    // strict mode: fp64 rows gathered by LDS-DMA, v_mfma_f64_16x16x4_f64 (k_missing.h, na_gram_lds_kernel); tail form for k = 16 j + 1, + 2
    const int nb = (nc + 3) / 4;
    const int ntail = h->k - 16 * (h->NKQ - 1);
    const bool tl = h->NKQ >= 2 && (ntail == 1 || ntail == 2);
#define NNLM_NAGL(N_, TL_) na_gram_lds_kernel<double, N_, TL_><<<nb, 256, 0, h->stream>>>(h->na_ptr[which], h->na_meta[which], h->na_idx[which], (const double *)h->Yrow, h->Graw, h->Gcols, c1, c0, h->k, upper_only ? 1 : 0)
    switch (h->NKQ) {
    case 1: NNLM_NAGL(1, false); break;
    case 2: if (tl) NNLM_NAGL(1, true); else NNLM_NAGL(2, false); break;
    case 3: if (tl) NNLM_NAGL(2, true); else NNLM_NAGL(3, false); break;
    default: if (tl) NNLM_NAGL(3, true); else NNLM_NAGL(4, false); break;
    }
#undef NNLM_NAGL
    return NNLM_OK;
}

// KL methods: no Gram, no cross product -- one block per column streams the column of A and the fixed factor.
// After a W half-step has been enqueued into the alternate buffers: make them current.
static void swap_w(nnlm_handle *h)
{
    h->wcur ^= 1;
    h->W64 = h->W64b[h->wcur];
    h->Wop = h->Wopb[h->wcur];
}

// Phases of a sharded half-step (test hooks drive virtual ranks phase by phase; production runs PH_ALL):
//   PH_A   cross product + Gram over this rank's contraction slab, folded into the [G | C] buffer (before the all-reduce)
//   PH_B   sweep of this rank's columns into the packed slab (after the all-reduce, before the all-gather)
//   PH_C   unpack of the all-gathered slabs into the resident layouts
enum { PH_ALL = 0, PH_A = 1, PH_B = 2, PH_C = 3 };
static int half_step_solve(nnlm_handle *h, int which, const double reg[3], unsigned inner_max_iter, double inner_rel_tol, int method,
                           int nslabs, bool speculative, int phase, bool colshard = false);

static int pack_prepare(nnlm_handle *h, int ncols, struct ShardCols *out, size_t tail);
static int pack_gather_unpack(nnlm_handle *h, int which, int phase);

// Columns of the factor being solved that this rank sweeps (multi-GPU): equal slabs of cpr columns (multiple of 256).
struct ShardCols {
    int cpr, col0, col1;
};
static ShardCols shard_cols(const nnlm_handle *h, int ncols)
{
    ShardCols c;
    c.cpr = round_up_i((ncols + h->nranks - 1) / h->nranks, 256); // (a whole number of cross-product tiles in both orientations)
    c.col0 = h->rank * c.cpr < ncols ? h->rank * c.cpr : ncols;
    c.col1 = c.col0 + c.cpr < ncols ? c.col0 + c.cpr : ncols;
    return c;
}

static int ensure_na_lists(nnlm_handle *h, int which, const uint32_t *bits, int words, int p, int ncols);

// Across GPUs the KL methods shard by columns of the factor being solved (the solvers need a column's whole contraction and
// exchange nothing while they run): a rank solves its columns into the packed slab, ONE all-gather returns the factor.
static int half_step_kl(nnlm_handle *h, int which, const double reg[3], unsigned inner_max_iter, double inner_rel_tol, int method,
                        bool speculative, int phase)
{
    h->pack_tail = 0; // (the KL slabs travel alone)
    if (h->sharded && phase == PH_A) return NNLM_OK; // (test hooks: nothing to all-reduce)
    if (h->sharded && phase == PH_C) {
        int rc = pack_gather_unpack(h, which, phase);
        if (rc == NNLM_OK && which == 0 && !speculative) swap_w(h);
        return rc;
    }
    KlArgs a;
    a.k = h->k;
    a.r0 = reg[0];
    a.r1 = reg[1];
    a.r2 = reg[2];
    a.max_iter = inner_max_iter;
    a.rel_tol = inner_rel_tol;
    a.sweeps = h->sweeps + (speculative ? (h->sw_active ^ 1) : h->sw_active);
    a.A = h->A;
    a.op_f64 = (h->prec == NNLM_PREC_F64) ? 1 : 0;
    if (which == 1) {
        a.X = h->H64; a.Xout = h->H64; a.ldx = h->mpad; a.Y = h->W64; a.ldy = h->npad;
        a.a_col_stride = (size_t)h->npad; a.a_i_stride = 1;
        a.bits = h->any_missing ? h->miss : nullptr; a.words = h->npad / 32;
        a.p = h->n; a.ncols = h->m;
        a.mask = h->has_hmask ? h->Hmask : nullptr;
        a.op = nullptr; a.op_mode = 0; a.op_ld = 0;
    } else {
        a.X = h->W64; a.Xout = h->W64b[h->wcur ^ 1]; a.ldx = h->npad; a.Y = h->H64; a.ldy = h->mpad;
        a.a_col_stride = 1; a.a_i_stride = (size_t)h->npad;
        a.bits = h->any_missing ? h->missT : nullptr; a.words = h->mpad / 32;
        a.p = h->m; a.ncols = h->n;
        a.mask = h->has_wmask ? h->Wmask : nullptr;
        a.op = h->Wopb[h->wcur ^ 1]; a.op_mode = (h->prec == NNLM_PREC_F64) ? 0 : 1; a.op_ld = h->npad;
    }
    const int ld_con = (which == 1) ? h->npad : h->mpad; // padded contraction length
    const int ncols_all = a.ncols;
    a.ldo = a.ldx;
    if (h->sharded) { // this rank's columns only, into the packed slab; the unpack writes masters and operands
        ShardCols sc;
        int rcp = pack_prepare(h, ncols_all, &sc, 0);
        if (rcp != NNLM_OK) return rcp;
        a.col0 = sc.col0;
        a.ncols = sc.col1;
        a.Xout = h->pack_send;
        a.ldo = sc.cpr;
        a.ocol0 = sc.col0;
        a.op = nullptr;
        a.op_mode = 0;
    }
    {
    ProfScope ps(h, which == 1 ? P_SWEEP_H : P_SWEEP_W);
    // The register-resident kernels start from the states y = Yt^T x of ALL columns (one GEMM into a matrix-sized buffer) and read
    // contraction-contiguous data (a transposed copy of A for the W half-step): one to two more copies of the matrix in HBM.  Where they
    // cannot be had, the streaming kernel takes the half-step in column chunks, with whatever scratch it can get (strided reads of A for
    // the W half-step: slow, and the reference's arithmetic either way) -- an allocation that fails here is not an error.
    bool tile_path = h->prec == NNLM_PREC_F32 && kl_tile_fits(a.p, h->k, a.mask ? h->MW : 0);
    bool reg64_path = h->prec == NNLM_PREC_F64 && !generic_rank(h) && kl64_fits(a.p, h->k);
    // No room for the matrix-sized What: with two row buffers (contractions up to 20480) kl_tile_kernel forms its starting states itself
    // from the rows of the fixed factor -- 8-12 % slower than the GEMM + read-back at config 3 (4.67 -> 5.03 ms per step,
    // profiles/r06_kl_own_init_ab.log), far ahead of the streaming kernel.  `what_tight` keeps a failed allocation from being retried in
    // every half-step; nnlm_set_matrix clears it.
    bool kl_own_init = false;
    if (tile_path && !h->What) {
        if (h->what_tight || big_malloc((void **)&h->What, (size_t)h->npad * h->mpad * sizeof(float) + 4096) != hipSuccess) {
            h->what_tight = true;
            if (kl_tile_nbuf(kl_tile_ept4(a.p)) == 2) kl_own_init = true;
            else tile_path = false;
        }
    }
    if (reg64_path && !h->What64 &&
        (h->what_tight || big_malloc((void **)&h->What64, (size_t)h->npad * h->mpad * sizeof(double) + 4096) != hipSuccess)) {
        h->what_tight = true;
        reg64_path = false;
    }
    if ((tile_path || reg64_path) && which == 0) {
        const int rc = h->at_tight ? NNLM_ERR_UNSUPPORTED : ensure_AT(h, true);
        if (rc == NNLM_ERR_UNSUPPORTED) {
            // the W half-step goes to the streaming kernel: the matrix-sized starting-state buffer is exactly the memory its scratch
            // needs (the H half-step then forms its states itself / streams too), and the failing allocation is not retried every half-step
            h->at_tight = true;
            tile_path = reg64_path = kl_own_init = false;
            hipFree(h->What), hipFree(h->What64);
            h->What = nullptr, h->What64 = nullptr;
            h->what_tight = true;
        } else if (rc != NNLM_OK) return rc;
    }
    if (tile_path) {
        // ---- fp32-operand mode: register-resident state, rows of the fixed factor staged through LDS (kl_tile_kernel) ----
        KlTileArgs ta;
        const size_t cnt = (size_t)h->KP * h->mpad; // fp32 [KP][mpad] copy of H (fixed factor of the W half-step, operand of the GEMM below)
        factor_to_f32_kernel<<<(unsigned)((cnt + 255) / 256), 256, 0, h->stream>>>(h->H64, cnt, h->Hkq);
        if (!h->klsw) HIPCHK(h, hipMalloc(&h->klsw, (size_t)h->KP * 8));
        // starting state vectors y = Yt^T x of ALL columns as one GEMM, in the layout the solver reads: [column][contraction]
        const int k2 = round_up_i(h->k, 2);
        const int lds = 2 * k2 * ERRF_TILE * (int)sizeof(float);
        set_dyn_lds((const void *)wh_store_kernel, lds, "wh_store_kernel");
        // (multi-GPU: only the column tiles of this rank's shard -- its first column is a multiple of 256)
        const int ct0 = a.col0 / ERRF_TILE, ct1 = (a.ncols + ERRF_TILE - 1) / ERRF_TILE;
        const int ny = ct1 > ct0 ? ct1 - ct0 : 0;
        const size_t cofs = (size_t)ct0 * ERRF_TILE;
        if (which == 1) {
            const int nx = h->npad / ERRF_TILE;
            if (ny > 0 && !kl_own_init)
                wh_store_kernel<<<8u * ((nx + 7) / 8) * ny, 256, lds, h->stream>>>((const float *)h->Wop, h->npad, h->Hkq + cofs, h->mpad, k2,
                                                                                     h->What + cofs * h->npad, h->npad, nx);
            ta.Adata = (const float *)h->A;
            ta.Yf = (const float *)h->Wop;
        } else { // roles swapped: What^T [row of A][column of A], next to the transposed fp32 copy of A
            int rc = ensure_AT(h);
            if (rc != NNLM_OK) return rc;
            const int nx = h->mpad / ERRF_TILE;
            if (ny > 0 && !kl_own_init)
                wh_store_kernel<<<8u * ((nx + 7) / 8) * ny, 256, lds, h->stream>>>(h->Hkq, h->mpad, (const float *)h->Wop + cofs, h->npad, k2,
                                                                                     h->What + cofs * h->mpad, h->mpad, nx);
            ta.Adata = (const float *)h->AT;
            ta.Yf = h->Hkq;
        }
        ta.lda = (size_t)ld_con;
        ta.Yinit = kl_own_init ? nullptr : h->What;
        ta.ldyf = ld_con;
        ta.p = a.p;
        ta.ncols = a.ncols;
        ta.k = h->k;
        ta.X = a.X;
        ta.Xout = a.Xout;
        ta.ldx = a.ldx;
        ta.colbase = a.col0;
        ta.ldo = a.ldo;
        ta.ocol0 = a.ocol0;
        kl_sumw_kernel<<<h->k, 256, 0, h->stream>>>(a.Y, a.ldy, a.p, h->klsw);
        ta.sumw = h->klsw;
        ta.sumw_cols = nullptr;
        ta.ldsw = h->KP;
        if (h->any_missing) { // row sums over each column's non-missing entries (src/update_with_missing.cpp:122,130)
            if (!h->Yrow) HIPCHK(h, hipMalloc(&h->Yrow, yrow_bytes(h)));
            if (!h->klsw_cols) HIPCHK(h, hipMalloc(&h->klsw_cols, (size_t)(h->n > h->m ? h->n : h->m) * h->KP * 8));
            int rc = ensure_na_lists(h, which, a.bits, a.words, a.p, ncols_all);
            if (rc != NNLM_OK) return rc;
            factor_rows_kernel<<<(a.p + 255) / 256, 256, 0, h->stream>>>(a.Y, a.ldy, a.p, h->KP, h->Yrow);
            if (a.ncols > a.col0) // (this rank's columns only)
                kl_sumw_cols_kernel<<<(a.ncols - a.col0 + 3) / 4, 256, 0, h->stream>>>(h->na_ptr[which] + a.col0, h->na_meta[which] + a.col0, h->na_idx[which],
                                                                                      h->Yrow, h->KP, h->k, h->klsw, h->klsw_cols + (size_t)a.col0 * h->KP, h->KP,
                                                                                      a.ncols - a.col0);
            ta.sumw_cols = h->klsw_cols;
        }
        ta.r0 = reg[0];
        ta.r1 = reg[1];
        ta.r2 = reg[2];
        ta.mask = a.mask;
        ta.mw = h->MW;
        ta.max_iter = inner_max_iter;
        ta.rel_tol = inner_rel_tol;
        ta.op = a.op;
        ta.op_mode = a.op_mode;
        ta.op_ld = a.op_ld;
        ta.sweeps = a.sweeps;
        launch_kl_tile(method, ta, h->stream);
        h->kl_form[which] = kl_own_init ? 1 : 0;
    } else if (reg64_path) {
        // ---- strict fp64 mode: register-resident fp64 state, the row of the fixed factor parked in LDS between the passes ----
        Kl64Args ka;
        if (!h->klsw) HIPCHK(h, hipMalloc(&h->klsw, (size_t)h->KP * 8));
        const int k4 = round_up_i(h->k, 4);
        // (multi-GPU: only the column tiles of this rank's shard -- its first column is a multiple of 256)
        const int ct0 = a.col0 / 64, ct1 = (a.ncols + 63) / 64;
        const int ny = ct1 > ct0 ? ct1 - ct0 : 0;
        const size_t cofs = (size_t)ct0 * 64;
        if (which == 1) { // What64[j][i], the layout of A
            dim3 grid(h->npad / 64, ny > 0 ? ny : 1);
            if (ny > 0)
                wh_store64_kernel<<<grid, 256, 0, h->stream>>>(h->W64, h->npad, h->H64 + cofs, h->mpad, k4, h->What64 + cofs * h->npad, (size_t)h->npad,
                                                               h->n, h->m - (int)cofs);
            ka.Adata = (const double *)h->A;
        } else { // roles swapped: What64^T [row of A][column of A], next to the transposed copy of A
            int rc = ensure_AT(h);
            if (rc != NNLM_OK) return rc;
            dim3 grid(h->mpad / 64, ny > 0 ? ny : 1);
            if (ny > 0)
                wh_store64_kernel<<<grid, 256, 0, h->stream>>>(h->H64, h->mpad, h->W64 + cofs, h->npad, k4, h->What64 + cofs * h->mpad, (size_t)h->mpad,
                                                               h->m, h->n - (int)cofs);
            ka.Adata = (const double *)h->AT;
        }
        ka.lda = (size_t)ld_con;
        ka.Yinit = h->What64;
        ka.Y = a.Y;
        ka.ldy = a.ldy;
        ka.p = a.p;
        ka.ncols = a.ncols;
        ka.k = h->k;
        ka.X = a.X;
        ka.Xout = a.Xout;
        ka.ldx = a.ldx;
        ka.colbase = a.col0;
        ka.ldo = a.ldo;
        ka.ocol0 = a.ocol0;
        kl_sumw_kernel<<<h->k, 256, 0, h->stream>>>(a.Y, a.ldy, a.p, h->klsw);
        ka.sumw = h->klsw;
        ka.sumw_cols = nullptr;
        ka.ldsw = h->KP;
        if (h->any_missing) { // row sums over each column's non-missing entries (src/update_with_missing.cpp:122,130)
            if (!h->Yrow) HIPCHK(h, hipMalloc(&h->Yrow, yrow_bytes(h)));
            if (!h->klsw_cols) HIPCHK(h, hipMalloc(&h->klsw_cols, (size_t)(h->n > h->m ? h->n : h->m) * h->KP * 8));
            int rc = ensure_na_lists(h, which, a.bits, a.words, a.p, ncols_all);
            if (rc != NNLM_OK) return rc;
            factor_rows_kernel<<<(a.p + 255) / 256, 256, 0, h->stream>>>(a.Y, a.ldy, a.p, h->KP, h->Yrow);
            if (a.ncols > a.col0) // (this rank's columns only)
                kl_sumw_cols_kernel<<<(a.ncols - a.col0 + 3) / 4, 256, 0, h->stream>>>(h->na_ptr[which] + a.col0, h->na_meta[which] + a.col0, h->na_idx[which],
                                                                                      h->Yrow, h->KP, h->k, h->klsw, h->klsw_cols + (size_t)a.col0 * h->KP, h->KP,
                                                                                      a.ncols - a.col0);
            ka.sumw_cols = h->klsw_cols;
        }
        ka.r0 = reg[0];
        ka.r1 = reg[1];
        ka.r2 = reg[2];
        ka.mask = a.mask;
        ka.max_iter = inner_max_iter;
        ka.rel_tol = inner_rel_tol;
        ka.op = a.op;
        ka.op_mode = a.op_mode;
        ka.op_ld = a.op_ld;
        ka.sweeps = a.sweeps;
        launch_kl64(method, ka, h->stream);
        h->kl_form[which] = 2;
    } else {
        // ---- no size limits: state vectors and data columns streamed from a scratch buffer (kl_stream_kernel), `chunk` columns at a
        // time: the whole range when the scratch for it can be had (two vectors per column), otherwise as many as fit
        const size_t per_col = (size_t)2 * ld_con * esize(h);
        const int range = a.ncols > a.col0 ? a.ncols - a.col0 : 0;
        int chunk = range;
        if (range > 0 && h->klst_bytes < (size_t)range * per_col) {
            // A smaller scratch is kept only once an allocation of the full size has actually failed for this matrix (`klst_tight`):
            // otherwise the two half-steps of an iteration, whose sizes differ slightly (n x mpad against m x npad), would have the
            // larger one split into a full launch + a tail launch of a few columns in every iteration.
            if (h->klst_tight && h->klst_bytes >= (size_t)64 * per_col && h->klst_bytes / per_col >= (size_t)range / 8)
                chunk = (int)(h->klst_bytes / per_col);
            else {
                hipFree(h->klst);
                h->klst = nullptr;
                h->klst_bytes = 0;
                for (;; chunk = (chunk + 1) / 2) {
                    if (big_malloc(&h->klst, (size_t)chunk * per_col) == hipSuccess) break;
                    h->klst_tight = true;
                    if (chunk <= 64) return fail(h, NNLM_ERR_HIP, "KL solver: no scratch for even %d columns (%zu bytes)", chunk, (size_t)chunk * per_col);
                }
                h->klst_bytes = (size_t)chunk * per_col;
            }
        }
        h->kl_form[which] = 3;
        const int col_end = a.ncols;
        for (int c0 = a.col0; c0 < col_end; c0 += chunk) {
            KlArgs ac = a;
            ac.col0 = c0;
            ac.ncols = (c0 + chunk < col_end) ? c0 + chunk : col_end;
            // (the kernel addresses the scratch by absolute column: the chunk's slots start at its first column)
            char *st = (char *)h->klst - (size_t)c0 * per_col;
            const int rcs = (h->prec == NNLM_PREC_F64) ? launch_kl_stream<double>(h, method, ac, h->MW, st, (size_t)ld_con, h->stream)
                                                       : launch_kl_stream<float>(h, method, ac, h->MW, st, (size_t)ld_con, h->stream);
            if (rcs != NNLM_OK) return rcs;
        }
    }
    }
    LAUNCHCHK(h);
    if (h->sharded) {
        if (phase == PH_B) return NNLM_OK; // test hooks: the caller gathers the slabs
        int rc = pack_gather_unpack(h, which, phase);
        if (rc != NNLM_OK) return rc;
    }
    if (which == 0 && !speculative) swap_w(h);
    return NNLM_OK;
}

static int errors_launch(nnlm_handle *h, hipStream_t st, bool with_sweeps, int fused_nb, bool need_pen);
// speculative (W half-step only): the result goes to the alternate W buffers and the alternate sweep counter and is
// NOT made current; the caller accepts it later with swap_w() / sw_active ^= 1, or simply drops it.
static int half_step(nnlm_handle *h, int which, const double reg[3], unsigned inner_max_iter, double inner_rel_tol, int method,
                     bool partial_only = false, bool speculative = false, int phase = PH_ALL)
{
    if (!h || !h->A || !h->W64) return fail(h, NNLM_ERR_ARG, "half_step: matrix and factors must be set first");
    if (method < 1 || method > 4) return fail(h, NNLM_ERR_ARG, "method must be 1..4 (got %d)", method);
    HIPCHK(h, hipSetDevice(h->device));
    g_attr_err = hipSuccess; // (a refusal latched by an earlier call on this thread that returned before its LAUNCHCHK is not this call's)
    const int sg_which = h->sg_which; // what the previous sweep left behind (any half-step rewrites a factor: reset first)
    const int sg_other = h->sg_other;
    h->sg_prev = sg_which;
    h->sg_which = h->sg_other = -1;
    h->sg_request = false;
    if (generic_rank(h) && h->sharded && method < 3 && !h->any_missing && !h->dense_cols)
        return fail(h, NNLM_ERR_UNSUPPORTED, "rank > %d across GPUs: only the column-sharded form (nnlm_comm_set_form(h, NNLM_FORM_COLS))", NNLM_KQ_MAX);
    if (partial_only) phase = PH_A;
    if (method >= 3) return half_step_kl(h, which, reg, inner_max_iter, inner_rel_tol, method, speculative, phase);
    // Missing values across GPUs: every column has a Gram of its own, so the column is the unit: a rank forms the
    // cross product of ITS columns over the whole contraction, their Grams, solves them, and ONE all-gather returns the factor --
    // no all-reduce.  Test hooks: phase 1 is empty, phase 2 computes and packs, phase 3 unpacks.
    // Dense square loss across GPUs, two forms.  "reduce" (north_star's wording): contraction-sharded [G | C], one
    // all-reduce, column-sharded sweep, one all-gather.  "cols": the column-sharded form of the NA / KL paths -- a rank forms the cross
    // product of ITS columns over the whole contraction (it holds all of A and, after the previous all-gather, all of the fixed
    // factor), the full Gram of the fixed factor (2 k^2 p flops, replicated), sweeps its columns, ONE all-gather.  Same HBM bytes per
    // rank (1/N of A either way), no all-reduce of the (KP^2 + KP cols) doubles: nnlm_comm_set_form, NNLM_FORM_COLS (default) | NNLM_FORM_REDUCE.
    // (read when the communicator is set up: nnlm_comm_init)
    const bool colshard = h->sharded && (h->any_missing || h->dense_cols);
    if (colshard && phase == PH_A) return NNLM_OK;
    if (phase == PH_C || (phase == PH_B && !colshard)) {
        const HalfPlan pp = plan_half(h, which, h->rank, h->nranks);
        return half_step_solve(h, which, reg, inner_max_iter, inner_rel_tol, method, pp.S, speculative, phase, colshard);
    }
    if (h->any_missing && !h->Gcols) { // NA path workspaces, on first use
        if (!h->Yrow) HIPCHK(h, hipMalloc(&h->Yrow, yrow_bytes(h)));
        HIPCHK(h, hipMalloc(&h->Gcols, (size_t)(h->n > h->m ? h->n : h->m) * h->KP * h->KP * 8));
    }
    HalfPlan p = plan_half(h, which, colshard ? 0 : h->rank, colshard ? 1 : h->nranks);
    if (colshard) { // own columns only, whole contraction
        const ShardCols sc = shard_cols(h, (which == 1) ? h->m : h->n);
        const int tile = XPROD_TN_BJ;
        p.col_off = sc.col0;
        p.tiles_x = (sc.col1 - sc.col0 + tile - 1) / tile;
        if (p.tiles_x > 0) split_plan(p.tiles_x, p.stage_end - p.stage_begin, &p.S, &p.sps);
        const size_t need = (size_t)p.S * h->KP * ((which == 1) ? h->mpad : h->npad);
        if (need > h->Cx_elems) { // (fewer tiles -> deeper split-K than the single-GPU plan the slabs were sized for)
            sync_all(h);
            hipFree(h->Cx);
            h->Cx = nullptr;
            h->Cx_elems = 0;
            HIPCHK(h, hipMalloc(&h->Cx, need * 8));
            HIPCHK(h, hipMemset(h->Cx, 0, need * 8));
            h->Cx_elems = need;
        }
    }
    // Dense SCD half-step of the split-fp16 mode on ONE stream.  The small kernels between a sweep and the next cross product
    // cost as much as they overlap (kernel timeline: 47 us from sweep end to cross product start, 24 us from its end to the
    // next sweep, a third of it cross-stream event latency), so they are fused instead: gram_partial also yields max|factor|
    // (no absmax pass, no memset), gram_reduce also writes the chain-wave constants of the strict sweep, and nothing waits on
    // another stream.
    h->pack_ready = false;
    if (h->x16 && !h->sharded && !h->any_missing && method == 1 && !generic_rank(h)) {
        // The one-wavefront sweep kernel (k_sweep_q.h) leaves max|x| and the Gram partial sums of the factor it solved -- the fixed
        // factor of the NEXT half-step -- behind, computed from its LDS image (+1 us per sweep): the kernels in front of the cross
        // product are then factor16_fold (split copy + sum of the workgroups' slabs) and the sweep's operand image, neither of
        // which needs a fence -- gram_partial (12 us) and the "last block" step of gram_reduce_consts (most of its 15 us:
        // __threadfence() is a cross-XCD cache write-back here) are gone.
        const bool fastsw = sweep_fast(h);
        if (fastsw && !h->sg_slabs) {
            const int big = h->n > h->m ? h->n : h->m;
            HIPCHK(h, hipMalloc(&h->sg_slabs, sg_slab_count(h) * h->KP * h->KP * 8));
        }
        unsigned *smax_w = fastsw ? h->maxbits + 4 + (h->sg_par ^ 1) : nullptr; // this half-step's sweep writes its max here
        h->sg_request = fastsw;
        if (fastsw && sg_which == (which == 1 ? 0 : 1)) {
            {
                ProfScope ps(h, P_GRAM, h->stream);
                SweepImg im;
                im.img = nullptr; // (fp32-operand mode: sweep_scd_f_kernel builds its operands from Graw itself -- no image)
                im.NB = (h->k + 3) / 4, im.NP = sweepq_np(im.NB, false), im.k = h->k, im.r0 = reg[0], im.r1 = reg[1];
                if (h->fuse_err && which == 0 && sg_other == 0) {
                    // trace iteration, max|W| still in the word W's sweep left it in: the split copies of H (rows and kq-contiguous) and of
                    // W (kq-contiguous) and the fold in one launch; the fused kernel clears that word for this half-step's sweep
                    const size_t cnt = (size_t)h->KP * h->mpad;
                    const unsigned nblk = (unsigned)(h->KP * h->KP / 64 + (cnt + 1023) / 1024 + h->mpad / 64 + h->npad / 64);
                    factor16_fold_err_kernel<<<nblk, 1024, 0, h->stream>>>(h->H64, h->mpad, h->m, h->W64, h->npad, h->n, h->k, h->KP, h->maxbits + 4 + h->sg_par,
                                                                           smax_w, h->scal_exp + 1, h->scal_exp + 2, h->Y16, h->H16c, h->W16c, h->sg_slabs,
                                                                           h->sg_nslabs, h->Graw, im);
                    h->y16_for = -1;
                    h->err_zero_word = smax_w;
                } else if (h->fuse_err) { // (the general routine)
                    prepare_factor16(h, which, h->maxbits + 4 + h->sg_par, smax_w, which == 0 && sg_other == 0);
                    gram_fold_kernel<<<h->KP * h->KP / 64, 1024, 0, h->stream>>>(h->sg_slabs, h->sg_nslabs, h->KP, h->Graw, im);
                } else {
                    const double *Ym = (which == 1) ? h->W64 : h->H64;
                    const int ldm = (which == 1) ? h->npad : h->mpad, lim = (which == 1) ? h->n : h->m;
                    const size_t cnt = (size_t)h->KP * ldm;
                    factor16_fold_kernel<<<(unsigned)(h->KP * h->KP / 64 + (cnt + 1023) / 1024), 1024, 0, h->stream>>>(
                        Ym, ldm, lim, h->k, h->KP, ldm, h->maxbits + 4 + h->sg_par, h->scal_exp + 1, h->Y16, smax_w, h->sg_slabs, h->sg_nslabs, h->Graw, im);
                }
                h->pack_ready = true;
            }
            {
                ProfScope ps(h, which == 1 ? P_XPROD_H : (h->fuse_err ? P_XPROD_W_ERR : P_XPROD_W));
                launch_xprod16(h, which, p);
            }
            if (speculative) HIPCHK(h, hipEventRecord(h->ev_xdone, h->stream));
            return half_step_solve(h, which, reg, inner_max_iter, inner_rel_tol, method, p.S, speculative, phase);
        }
        unsigned *mb = h->maxbits + 1 + h->mb_par, *mb_next = h->maxbits + 1 + (h->mb_par ^ 1);
        h->mb_par ^= 1;
        const double *Ym = (which == 1) ? h->W64 : h->H64;
        const int ldm = (which == 1) ? h->npad : h->mpad, lim = (which == 1) ? h->n : h->m;
        int nb = (lim + GRAM_COLS_PER_BLOCK - 1) / GRAM_COLS_PER_BLOCK;
        if (nb < 1) nb = 1;
        {
            ProfScope ps(h, P_GRAM, h->stream);
            switch (h->NKQ) {
            case 1: gram_partial_kernel<1><<<nb, 256, 0, h->stream>>>(Ym, ldm, 0, lim, h->gslabs, mb); break;
            case 2: gram_partial_kernel<2><<<nb, 256, 0, h->stream>>>(Ym, ldm, 0, lim, h->gslabs, mb); break;
            case 3: gram_partial_kernel<3><<<nb, 256, 0, h->stream>>>(Ym, ldm, 0, lim, h->gslabs, mb); break;
            default: gram_partial_kernel<4><<<nb, 256, 0, h->stream>>>(Ym, ldm, 0, lim, h->gslabs, mb); break;
            }
            prepare_factor16(h, which, mb, smax_w);
            gram_reduce_kernel<<<(h->KP * h->KP + 255) / 256, 256, 0, h->stream>>>(h->gslabs, nb, h->KP, h->Graw);
            HIPCHK(h, hipMemsetAsync(mb_next, 0, sizeof(unsigned), h->stream)); // the max word the NEXT half-step's gram_partial accumulates into
        }
        {
            ProfScope ps(h, which == 1 ? P_XPROD_H : (h->fuse_err ? P_XPROD_W_ERR : P_XPROD_W));
            launch_xprod16(h, which, p);
        }
        if (speculative) HIPCHK(h, hipEventRecord(h->ev_xdone, h->stream));
        return half_step_solve(h, which, reg, inner_max_iter, inner_rel_tol, method, p.S, speculative, phase);
    }
    // 1. Gram of the fixed factor over this rank's contraction slab -- on the main stream, AHEAD of the cross product.  (Rounds 1-2
    // ran it on a second stream next to the A-streaming kernel: the cross product owns every CU's LDS, the Gram kernel got slots
    // only as its workgroups retired and stretched from 0.03 to 0.5-1.0 ms while slowing the cross product down with it --
    // profiles/r03_f64_cfg2_a_kernel_stats.csv: xprod_tn 0.44 .. 1.6 ms, gram_partial 0.03 .. 0.99 ms.)
    int gslabs = 0;
    // (multi-GPU, dense SCD, column form: the unpack of the previous half-step already summed the ranks' Gram partial sums of this
    //  half-step's fixed factor into Graw -- shard_gram_sum_kernel -- instead of every rank recomputing the whole Gram)
    const bool gram_cached = colshard && method == 1 && !h->any_missing && !generic_rank(h) && h->gshard_for == ((which == 1) ? 0 : 1);
    // (one GPU, strict mode, dense SCD: the sweep that solved the fixed factor left the Gram partial sums of its workgroups behind, as
    //  in the split-fp16 flow above -- one fold instead of gram_partial + gram_reduce over the whole factor)
    const bool sg_strict = !h->sharded && !h->x16 && h->prec == NNLM_PREC_F64 && method == 1 && !h->any_missing && !generic_rank(h);
    if (sg_strict) {
        if (!h->sg_slabs) {
            const int big = h->n > h->m ? h->n : h->m;
            HIPCHK(h, hipMalloc(&h->sg_slabs, sg_slab_count(h) * h->KP * h->KP * 8));
        }
        h->sg_request = true; // this half-step's sweep leaves its slabs for the next one
    }
    unsigned *gmb = nullptr;
    if (sg_strict && sg_which == ((which == 1) ? 0 : 1)) {
        ProfScope ps(h, P_GRAM, h->stream);
        SweepImg im; // (the fold also writes the sweep's operand image: no sweepq_pack_kernel launch)
        im.img = h->sweepq_img, im.NB = (h->k + 3) / 4, im.NP = sweepq_np(im.NB, true), im.k = h->k, im.r0 = reg[0], im.r1 = reg[1];
        gram_fold_kernel<<<h->KP * h->KP / 64, 1024, 0, h->stream>>>(h->sg_slabs, h->sg_nslabs, h->KP, h->Graw, im);
        h->pack_ready = true;
        h->gshard_for = -1;
    } else if (!gram_cached) {
        h->gshard_for = -1;
        ProfScope ps(h, P_GRAM, h->stream);
        const int CE = stage_elems(h, which);
        int c0 = p.stage_begin * CE, c1 = p.stage_end * CE;
        const int lim = (which == 1) ? h->n : h->m;
        if (c1 > lim) c1 = lim;
        if (c0 > c1) c0 = c1;
        // (one GPU, split-fp16 mode: the Gram pass also leaves max|fixed factor| for the split copies -- no absmax pass; the two
        // words alternate as in the dense path above, each zeroed one half-step before its use)
        if (h->x16 && !h->sharded && !generic_rank(h) && c0 == 0 && c1 == lim) {
            gmb = h->maxbits + 1 + h->mb_par;
            h->mb_par ^= 1;
        }
        if (which == 1) launch_gram(h, h->W64, h->npad, c0, c1, &gslabs, gmb);
        else launch_gram(h, h->H64, h->mpad, c0, c1, &gslabs, gmb);
        if (gmb) HIPCHK(h, hipMemsetAsync(h->maxbits + 1 + h->mb_par, 0, sizeof(unsigned), h->stream));
    }
    // 2. cross product slabs
    if (h->x16) {
        // (multi-GPU: the unpack of the previous half-step left max|fixed factor| behind -- no absmax pass)
        const int solved_by = (which == 1) ? 0 : 1; // the half-step that solved this half-step's fixed factor
        unsigned *mb = (h->sharded && h->upk_max_for == solved_by) ? h->maxbits + 6 + solved_by : gmb;
        // (dense SCD, column form: that unpack also wrote the split copy and its exponent -- the ranks' maxima travelled with the slabs)
        if (!(h->sharded && mb && h->y16_for == solved_by && !h->fuse_err)) prepare_factor16(h, which, mb);
        h->fixed_maxw = mb ? mb : h->maxbits;
    }
    if (p.tiles_x > 0) {
        ProfScope ps(h, which == 1 ? P_XPROD_H : (h->fuse_err ? P_XPROD_W_ERR : P_XPROD_W));
        if (generic_rank(h)) {
            int rcx = launch_xprod_generic(h, which, p);
            if (rcx != NNLM_OK) return rcx;
        } else if (h->x16) launch_xprod16(h, which, p);
        else {
            if (which == 0) {
                int rca = ensure_AT(h);
                if (rca != NNLM_OK) return rca;
            }
            launch_xprod_nkq<double>(h, which, p);
        }
    }
    if (speculative) HIPCHK(h, hipEventRecord(h->ev_xdone, h->stream)); // the error block starts once A is no longer streamed
    if (speculative && h->err_hook) { // (multi-GPU trace iteration: see err_hook)
        h->err_hook = false;
        int fnb = h->fused_nb;
        if (h->fuse_err && fnb == 0) { // a rank without columns launched no fused cross product: it contributes zeros to the two sums
            HIPCHK(h, hipMemsetAsync(h->partials, 0, 2 * sizeof(double), h->stream));
            HIPCHK(h, hipEventRecord(h->ev_xdone, h->stream));
            fnb = 1;
        }
        const int rce = errors_launch(h, h->stream_e, true, fnb, h->err_need_pen);
        if (rce != NNLM_OK) return rce;
        h->fused_nb = 0;
        h->err_launched = true;
    }
    // (multi-GPU) fold the split-K slabs into the contiguous [G | C] buffer; ONE all-reduce sums it over ranks
    if (h->sharded && !colshard) {
        const int ld = (which == 1) ? h->mpad : h->npad;
        const size_t cnt = (size_t)h->KP * ld;
        slab_reduce_kernel<<<(unsigned)((cnt + 255) / 256), 256, 0, h->stream>>>(h->Cx, p.S, cnt, h->red + (size_t)h->KP * h->KP);
        if (phase == PH_A) return NNLM_OK; // test hooks: the caller performs the exchange
        if (h->comm) {
            ProfScope ps(h, P_ALLREDUCE);
            ncclResult_t r = g_rccl.AllReduce(h->red, h->red, (size_t)h->KP * h->KP + cnt, ncclDouble, ncclSum, (ncclComm_t)h->comm, h->stream);
            if (r != ncclSuccess) return fail(h, NNLM_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
        }
    }
    return half_step_solve(h, which, reg, inner_max_iter, inner_rel_tol, method, p.S, speculative, phase, colshard);
}

static int shard_unpack(nnlm_handle *h, int which)
{
    ProfScope ps(h, P_UNPACK);
    const int ncols = (which == 1) ? h->m : h->n;
    const ShardCols sc = shard_cols(h, ncols);
    const int f64 = (h->prec == NNLM_PREC_F64) ? 1 : 0;
    unsigned *maxw = h->x16 ? h->maxbits + 6 + which : nullptr; // max|factor| for the next half-step's split copy (no absmax pass there)
    // dense SCD in the column form (pack_tail = KP * KP Gram partial sums + the rank's max|x|): max|factor| is known before an entry is
    // read, so the unpack writes the split-fp16 copy of the factor -- the fixed factor of the NEXT half-step -- and its exponent too
    const bool tail_max = h->pack_tail != 0 && h->x16;
    const size_t tail_max_off = tail_max ? (size_t)h->k * sc.cpr + (size_t)h->KP * h->KP : (size_t)-1;
    uint32_t *y16 = (tail_max && !generic_rank(h)) ? h->Y16 : nullptr;
    const int rows = y16 ? h->KP : h->k; // (the k meaningful rows of every rank's [KP][cpr] slab travel, not the padding; the split copy is zero padded)
    const size_t tot = (size_t)h->nranks * rows * sc.cpr;
    if (maxw && !tail_max) HIPCHK(h, hipMemsetAsync(maxw, 0, sizeof(unsigned), h->stream));
    h->upk_max_for = maxw ? which : -1;
    const size_t rank_stride = (size_t)h->k * sc.cpr + h->pack_tail;
    if (which == 1)
        shard_unpack_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, h->stream>>>(h->pack_all, h->nranks, rows, sc.cpr, h->k, ncols, h->H64,
                                                                                   h->mpad, nullptr, 0, 0, f64, maxw, rank_stride, tail_max_off, y16,
                                                                                   h->mpad, h->scal_exp + 1);
    else
        shard_unpack_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, h->stream>>>(h->pack_all, h->nranks, rows, sc.cpr, h->k, ncols,
                                                                                   h->W64b[h->wcur ^ 1], h->npad, h->Wopb[h->wcur ^ 1],
                                                                                   f64 ? 0 : 1, h->npad, f64, maxw, rank_stride, tail_max_off, y16,
                                                                                   h->npad, h->scal_exp + 1);
    h->y16_for = y16 ? which : -1;
    if (h->gshard_for == which) h->gshard_for = -1; // (that factor has just been rewritten)
    if (h->pack_tail) { // the Gram of the factor just gathered = the sum of the ranks' partial sums, in rank order
        const int cnt = h->KP * h->KP;
        shard_gram_sum_kernel<<<(cnt + 255) / 256, 256, 0, h->stream>>>(h->pack_all, h->nranks, rank_stride, (size_t)h->k * sc.cpr, cnt, h->Graw);
        h->gshard_for = which;
    }
    LAUNCHCHK(h);
    return NNLM_OK;
}

// packed slab [KP][cpr] this rank solves its columns into (zeroed) and the gathered [nranks][KP][cpr]
// tail: doubles that travel behind the slab's k meaningful rows (at offset k * cpr: the slab's padding rows are not gathered)
static int pack_prepare(nnlm_handle *h, int ncols, struct ShardCols *out, size_t tail)
{
    const ShardCols sc = shard_cols(h, ncols);
    const size_t need = (size_t)h->KP * sc.cpr + tail;
    if (h->pack_elems < need * h->nranks) {
        hipFree(h->pack_send);
        hipFree(h->pack_all);
        h->pack_send = h->pack_all = nullptr;
        h->pack_elems = 0;
        HIPCHK(h, hipMalloc(&h->pack_send, need * 8));
        HIPCHK(h, hipMalloc(&h->pack_all, need * h->nranks * 8));
        h->pack_elems = need * h->nranks;
        HIPCHK(h, hipMemsetAsync(h->pack_send, 0, need * 8, h->stream));
    } else if (sc.col1 <= sc.col0) // a rank without columns sends zeros (nothing below writes its slab or its tail)
        HIPCHK(h, hipMemsetAsync(h->pack_send, 0, need * 8, h->stream));
    // (otherwise no memset per half-step: the solvers write every live entry -- rows < k of the rank's columns --, the tail is written
    //  whole, and the unpack reads nothing else)
    *out = sc;
    return NNLM_OK;
}
// ONE ncclAllGather of the packed slabs, then every rank writes masters and GEMM operands of all columns
static int pack_gather_unpack(nnlm_handle *h, int which, int phase)
{
    const int ncols = (which == 1) ? h->m : h->n;
    const ShardCols sc = shard_cols(h, ncols);
    if (h->comm) {
        ProfScope ps(h, P_ALLGATHER);
        ncclResult_t r = g_rccl.AllGather(h->pack_send, h->pack_all, (size_t)h->k * sc.cpr + h->pack_tail, ncclDouble, (ncclComm_t)h->comm,
                                          h->stream); // rows 0 .. k-1 (+ the Gram partial sums behind them)
        if (r != ncclSuccess) return fail(h, NNLM_ERR_COMM, "ncclAllGather failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    } else if (phase == PH_ALL)
        return fail(h, NNLM_ERR_COMM, "virtual rank %d of %d has no communicator: drive it with nnlm_debug_phase()", h->rank, h->nranks);
    return shard_unpack(h, which);
}

// 3. per-column solve (+ multi-GPU: all-gather of the solved column slabs and unpack into the resident layouts)
// colshard: the cross product was formed over the whole contraction for this rank's columns only (missing values: per-column
// Grams make the column the natural unit) -- its split-K slabs are read directly, nothing was all-reduced.
static int half_step_solve(nnlm_handle *h, int which, const double reg[3], unsigned inner_max_iter, double inner_rel_tol, int method,
                           int nslabs, bool speculative, int phase, bool colshard)
{
    const int ncols = (which == 1) ? h->m : h->n;
    h->cur_which = which;
    // dense SCD in the column form: the Gram partial sums of this rank's columns travel with its slab (shard_gram_sum_kernel)
    // (+ one double: the rank's max|x|, so that the unpack can write the split-fp16 copy of the factor as well)
    h->pack_tail = (h->sharded && colshard && method == 1 && !h->any_missing && !generic_rank(h)) ? (size_t)h->KP * h->KP + 1 : 0;
    if (phase != PH_C) {
        ProfScope ps(h, which == 1 ? P_SWEEP_H : P_SWEEP_W);
        SweepArgs a;
        a.Graw = h->Graw;
        a.KPg = h->KP;
        a.Cx = (h->sharded && !colshard) ? h->red + (size_t)h->KP * h->KP : h->Cx;
        a.nslabs = (h->sharded && !colshard) ? 1 : nslabs;
        a.k = h->k;
        a.r0 = reg[0];
        a.r1 = reg[1];
        a.r2 = reg[2];
        a.max_iter = inner_max_iter;
        a.rel_tol = inner_rel_tol;
        a.sweeps = h->sweeps + (speculative ? (h->sw_active ^ 1) : h->sw_active);
        a.col0 = 0;
        a.ocol0 = 0;
        const bool sg = h->sg_request; // (only on the dense one-GPU split-fp16 path, where k_sweep_q.h runs)
        h->sg_request = false;
        if (sg) {
            a.maxbits = h->x16 ? h->maxbits + 4 + (h->sg_par ^ 1) : nullptr; // (max|x| is the split copy's scale: fp32-operand mode only)
            a.gram_slabs = h->sg_slabs;
        }
        if (which == 1) {
            a.X = h->H64;
            a.Xout = h->H64;
            a.ldx = a.ldo = h->mpad;
            a.ldc = h->mpad;
            a.slab_stride = (size_t)h->KP * h->mpad;
            a.ncols = h->m;
            a.mask = h->has_hmask ? h->Hmask : nullptr;
            a.op = nullptr; // (H has no GEMM-operand copy: every consumer reads the master or makes its own split / fp32 copy)
            a.op_mode = 0;
            a.op_ld = 0;
        } else {
            a.X = h->W64;
            a.Xout = h->W64b[h->wcur ^ 1];
            a.ldx = a.ldo = h->npad;
            a.ldc = h->npad;
            a.slab_stride = (size_t)h->KP * h->npad;
            a.ncols = h->n;
            a.mask = h->has_wmask ? h->Wmask : nullptr;
            a.op = h->Wopb[h->wcur ^ 1];
            a.op_mode = (h->prec == NNLM_PREC_F64) ? 0 : 1;
            a.op_ld = h->npad;
        }
        a.op_f64 = (h->prec == NNLM_PREC_F64) ? 1 : 0;
        if (h->sharded) { // sweep only this rank's columns into the packed slab; the unpack writes masters and operands
            ShardCols sc;
            int rcp = pack_prepare(h, ncols, &sc, h->pack_tail);
            if (rcp != NNLM_OK) return rcp;
            a.col0 = sc.col0;
            a.ncols = sc.col1;
            a.Xout = h->pack_send;
            a.ldo = sc.cpr;
            a.ocol0 = sc.col0;
            a.op = nullptr;
            a.op_mode = 0;
            if (h->pack_tail) {
                if (!h->sg_slabs) {
                    const int big = h->n > h->m ? h->n : h->m;
                    HIPCHK(h, hipMalloc(&h->sg_slabs, sg_slab_count(h) * h->KP * h->KP * 8));
                }
                a.gram_slabs = h->sg_slabs; // (one slab per workgroup of sweep_scd_q_kernel; folded behind the packed slab below)
                a.maxbits = h->x16 ? h->maxbits + 8 : nullptr; // (the sweep's atomicMax; handed to the tail and cleared by gram_fold_tail_kernel)
            }
        }
        if (h->any_missing) {
            // per-column Gram over the finite rows of each column (src/update_with_missing.cpp:90), then the solver
            const int p_len = (which == 1) ? h->n : h->m;
            {
                int rcg = launch_na_gram(h, which, which == 1 ? h->miss : h->missT, (which == 1 ? h->npad : h->mpad) / 32, p_len, ncols, a.col0, a.ncols, !generic_rank(h));
                if (rcg != NNLM_OK) return rcg;
            }
            a.Graw = h->Gcols;
            a.g_upper = generic_rank(h) ? 0 : 1; // (what launch_na_gram was told to write: ranks <= 64 keep the upper triangle only)
            if (generic_rank(h)) {
                int rcs = launch_sweep_generic(h, method, a, (size_t)h->KP * h->KP);
                if (rcs != NNLM_OK) return rcs;
            } else
                launch_colsolve(h, method, a, (size_t)h->KP * h->KP);
        } else if (a.ncols > a.col0) {
            int rcs = launch_sweep(h, method, a);
            if (rcs != NNLM_OK) return rcs;
            if (h->sharded && h->pack_tail) {
                const int nsl = h->sweep_wgs; // (one slab per workgroup of the sweep launch)
                gram_fold_tail_kernel<<<h->KP * h->KP / 64, 1024, 0, h->stream>>>(h->sg_slabs, nsl, h->KP, h->pack_send + (size_t)h->k * a.ldo, h->maxbits + 8);
            }
            if (sg) { // the next half-step finds max and Gram partial sums of this factor
                h->sg_nslabs = h->sweep_wgs; // (one slab per workgroup of the sweep launch)
                h->sg_par ^= 1;
                h->sg_which = which;
                h->sg_other = h->sg_prev; // the word this sweep did not touch
            }
        }
        LAUNCHCHK(h);
        if (h->sharded && phase == PH_B) return NNLM_OK; // test hooks: the caller gathers the slabs
    }
    if (h->sharded) {
        int rc = pack_gather_unpack(h, which, phase);
        if (rc != NNLM_OK) return rc;
    }
    if (which == 0 && !speculative) swap_w(h);
    return NNLM_OK;
}

extern "C" int nnlm_half_step(nnlm_handle *h, int which, const double reg[3], unsigned inner_max_iter, double inner_rel_tol, int method)
{
    if (which != 0 && which != 1) return fail(h, NNLM_ERR_ARG, "nnlm_half_step: which must be 0 (W) or 1 (H)");
    if (!reg) return fail(h, NNLM_ERR_ARG, "nnlm_half_step: reg is NULL");
    return half_step(h, which, reg, inner_max_iter, inner_rel_tol, method);
}

// Test hook: run the cross-product + Gram stage of a half-step for this (virtual) rank and return the partial
// [G (k x k) | C (k x cols)] sums, column-major, before any all-reduce and before the regularisation edits.
extern "C" int nnlm_debug_partial(nnlm_handle *h, int which, double *G_out, double *C_out)
{
    if (!h || !h->W64) return fail(h, NNLM_ERR_ARG, "nnlm_debug_partial: matrix and factors must be set first");
    if (which != 0 && which != 1) return fail(h, NNLM_ERR_ARG, "nnlm_debug_partial: which must be 0 or 1");
    const double z[3] = {0, 0, 0};
    // sharded handles return right after the slab fold; a 1-rank handle runs a zero-sweep solve (a no-op on the
    // factors) and its slabs are folded here
    int rc = half_step(h, which, z, 0, 0.0, 1, true);
    if (rc != NNLM_OK) return rc;
    const int ld = (which == 1) ? h->mpad : h->npad;
    if (!h->sharded) {
        const HalfPlan q = plan_half(h, which, 0, 1);
        const size_t cnt = (size_t)h->KP * ld;
        slab_reduce_kernel<<<(unsigned)((cnt + 255) / 256), 256, 0, h->stream>>>(h->Cx, q.S, cnt, h->red + (size_t)h->KP * h->KP);
    }
    sync_all(h);
    const int KP = h->KP, k = h->k;
    const int cols = (which == 1) ? h->m : h->n;
    std::vector<double> buf((size_t)KP * KP + (size_t)KP * ld);
    HIPCHK(h, hipMemcpy(buf.data(), h->red, buf.size() * 8, hipMemcpyDeviceToHost));
    if (G_out)
        for (int r = 0; r < k; r++)
            for (int q2 = 0; q2 < k; q2++) G_out[(size_t)r * k + q2] = buf[(size_t)q2 * KP + r];
    if (C_out)
        for (int c = 0; c < cols; c++)
            for (int q2 = 0; q2 < k; q2++) C_out[(size_t)c * k + q2] = buf[(size_t)KP * KP + (size_t)q2 * ld + c];
    return NNLM_OK;
}

// Test hooks for virtual ranks (several handles in one process, no communicator): run ONE phase of a sharded half-step
// (1 = PH_A, 2 = PH_B, 3 = PH_C) and let nnlm_debug_exchange() stand in for the collective between phases.
extern "C" int nnlm_debug_phase(nnlm_handle *h, int which, int phase, const double reg[3], unsigned inner_max_iter, double inner_rel_tol,
                                int method)
{
    if (!h || !reg || (which != 0 && which != 1) || phase < PH_A || phase > PH_C) return fail(h, NNLM_ERR_ARG, "nnlm_debug_phase: bad arguments");
    if (!h->sharded) return fail(h, NNLM_ERR_ARG, "nnlm_debug_phase: handle is not sharded (call nnlm_comm_init first)");
    // (nothing is reset here: nnlm_set_factors -- the only other writer of the factors -- invalidates what the last unpack left behind,
    //  so virtual ranks take the production path: cached Gram, max|factor| and split copy from the previous half-step's unpack)
    return half_step(h, which, reg, inner_max_iter, inner_rel_tol, method, false, false, phase);
}

// stage 1: what ncclAllReduce does to the [G | C] buffers (host sum in rank order); stage 2: what ncclAllGather does to the
// packed column slabs.  hs[r] must be the handle of virtual rank r; all handles describe the same problem.
extern "C" int nnlm_debug_exchange(nnlm_handle **hs, int P, int which, int stage)
{
    if (!hs || P < 1 || (which != 0 && which != 1) || (stage != 1 && stage != 2)) return fail(nullptr, NNLM_ERR_ARG, "nnlm_debug_exchange: bad arguments");
    nnlm_handle *h0 = hs[0];
    for (int r = 0; r < P; r++) {
        if (!hs[r] || hs[r]->nranks != P || hs[r]->rank != r) return fail(h0, NNLM_ERR_ARG, "nnlm_debug_exchange: hs[%d] is not virtual rank %d of %d", r, r, P);
        HIPCHK(hs[r], hipSetDevice(hs[r]->device));
        sync_all(hs[r]);
    }
    if (stage == 1) {
        const int ld = (which == 1) ? h0->mpad : h0->npad;
        const size_t cnt = (size_t)h0->KP * h0->KP + (size_t)h0->KP * ld;
        std::vector<double> sum(cnt, 0.0), tmp(cnt);
        for (int r = 0; r < P; r++) {
            HIPCHK(hs[r], hipMemcpy(tmp.data(), hs[r]->red, cnt * 8, hipMemcpyDeviceToHost));
            for (size_t e = 0; e < cnt; e++) sum[e] += tmp[e];
        }
        for (int r = 0; r < P; r++) HIPCHK(hs[r], hipMemcpy(hs[r]->red, sum.data(), cnt * 8, hipMemcpyHostToDevice));
    } else {
        const int ncols = (which == 1) ? h0->m : h0->n;
        const ShardCols sc = shard_cols(h0, ncols);
        const size_t per = (size_t)h0->k * sc.cpr + h0->pack_tail; // (as the ncclAllGather call: the first k rows of the slab + its tail)
        std::vector<double> all(per * P);
        for (int r = 0; r < P; r++) HIPCHK(hs[r], hipMemcpy(all.data() + per * r, hs[r]->pack_send, per * 8, hipMemcpyDeviceToHost));
        for (int r = 0; r < P; r++) HIPCHK(hs[r], hipMemcpy(hs[r]->pack_all, all.data(), per * P * 8, hipMemcpyHostToDevice));
    }
    return NNLM_OK;
}

extern "C" int nnlm_iterate(nnlm_handle *h, unsigned n_iter, const double alpha[3], const double beta[3], unsigned inner_max_iter,
                            double inner_rel_tol, int method)
{
    for (unsigned i = 0; i < n_iter; i++) {
        int rc = half_step(h, 0, alpha, inner_max_iter, inner_rel_tol, method); // update W, src/nnmf.cpp:131
        if (rc != NNLM_OK) return rc;
        rc = half_step(h, 1, beta, inner_max_iter, inner_rel_tol, method); // update H, src/nnmf.cpp:133
        if (rc != NNLM_OK) return rc;
    }
    return NNLM_OK;
}

extern "C" int nnlm_take_sweeps(nnlm_handle *h, long long *sweeps, int reset)
{
    if (!h) return fail(nullptr, NNLM_ERR_ARG, "nnlm_take_sweeps: handle is NULL");
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long v = 0;
    const unsigned long long *src = h->sweeps + h->sw_active;
    if (h->sharded && h->comm) {
        HIPCHK(h, hipMemcpyAsync(h->sweeps_tmp, src, sizeof v, hipMemcpyDeviceToDevice, h->stream));
        ncclResult_t r = g_rccl.AllReduce(h->sweeps_tmp, h->sweeps_tmp, 1, ncclUint64, ncclSum, (ncclComm_t)h->comm, h->stream);
        if (r != ncclSuccess) return fail(h, NNLM_ERR_COMM, "ncclAllReduce (sweep counter) failed");
        src = h->sweeps_tmp;
    }
    HIPCHK(h, hipMemcpyAsync(&v, src, sizeof v, hipMemcpyDeviceToHost, h->stream));
    if (reset) HIPCHK(h, hipMemsetAsync(h->sweeps + h->sw_active, 0, sizeof v, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (sweeps) *sweeps = (long long)v;
    return NNLM_OK;
}

extern "C" int nnlm_sync(nnlm_handle *h)
{
    if (!h) return fail(nullptr, NNLM_ERR_ARG, "nnlm_sync: handle is NULL");
    HIPCHK(h, hipSetDevice(h->device));
    sync_all(h);
    LAUNCHCHK(h);
    return NNLM_OK;
}

// ---------------------------------------------------------------------------------------------
// error block
// ---------------------------------------------------------------------------------------------
// Enqueue the error block on stream `st` for the CURRENT factors (pointers captured now); the 8 sums and, when
// `with_sweeps`, the active sweep counter land in the pinned host_res[0..8]; ev_err fires when they are there.
// fused_nb > 0: the two sums were left as fused_nb partial pairs in h->partials by the cross product of the speculative
// W half-step (xprod16_err_kernel); only their reduction, the penalties and the counters remain.
// need_pen: the six penalty sums of add_penalty() -- nnlm_run skips them when alpha = beta = 0 (penalties() then uses none of them)
static int errors_launch(nnlm_handle *h, hipStream_t st, bool with_sweeps, int fused_nb, bool need_pen)
{
    const int k4 = round_up_i(h->k, 4);
    if (fused_nb > 0) {
        HIPCHK(h, hipStreamWaitEvent(st, h->ev_xdone, 0)); // (ahead of the scope: the wait for the fused cross product is not this phase's time)
        ProfScope ps(h, P_ERR_REDUCE, st);
        reduce_partials_kernel<<<1, REDUCE_THREADS, 0, st>>>(h->partials, (size_t)fused_nb, 2, h->scal);
        if (h->sharded && h->comm) { // every rank's fused cross product covered ITS part of A: sum the two sums
            ncclResult_t r = g_rccl.AllReduce(h->scal, h->scal, 2, ncclDouble, ncclSum, (ncclComm_t)h->comm, st);
            if (r != ncclSuccess) return fail(h, NNLM_ERR_COMM, "ncclAllReduce (fused error sums) failed");
        }
    } else {
        ProfScope ps(h, P_ERRORS, st);
        const uint32_t *miss = h->any_missing ? h->miss : nullptr;
        size_t nb;
        // multi-GPU: each rank reduces its share of the j-tiles; the two sums are all-reduced below
        const bool err_generic = generic_rank(h) && h->prec == NNLM_PREC_F32; // rank > 64: the LDS-free kernel with fp32 A
        const int tile = (h->prec == NNLM_PREC_F64 || err_generic) ? ERR_TILE : ERRF_TILE;
        const int tj = h->mpad / tile, per = (tj + h->nranks - 1) / h->nranks;
        const int jt0 = h->sharded ? (h->rank * per < tj ? h->rank * per : tj) : 0;
        const int jcnt = h->sharded ? ((jt0 + per < tj ? jt0 + per : tj) - jt0) : tj;
        if (jcnt <= 0) {
            nb = 0;
        } else if (h->prec == NNLM_PREC_F64 && !generic_rank(h)) {
            // one 64-row i-tile x a chunk of j-tiles per block; ~8 rounds of the 2 x 256 resident blocks
            const int nit = h->npad / ERR_TILE;
            int nchunks = (8 * 512 + nit / 2) / nit;
            if (nchunks < 1) nchunks = 1;
            if (nchunks > jcnt) nchunks = jcnt;
            const int chunk = (jcnt + nchunks - 1) / nchunks;
            nchunks = (jcnt + chunk - 1) / chunk;
            nb = (size_t)nit * nchunks;
            const int lds = errors64_lds_bytes(k4);
            if (miss) {
                set_dyn_lds((const void *)errors64_kernel<true>, lds, "errors64_kernel");
                errors64_kernel<true><<<(unsigned)nb, ERR64_THREADS, lds, st>>>((const double *)h->A, h->npad, miss, h->W64, h->npad, h->H64, h->mpad, h->n,
                                                                                h->m, k4, h->partials, jt0, jcnt, chunk, nit);
            } else {
                set_dyn_lds((const void *)errors64_kernel<false>, lds, "errors64_kernel");
                errors64_kernel<false><<<(unsigned)nb, ERR64_THREADS, lds, st>>>((const double *)h->A, h->npad, nullptr, h->W64, h->npad, h->H64, h->mpad,
                                                                                 h->n, h->m, k4, h->partials, jt0, jcnt, chunk, nit);
            }
        } else if (h->prec == NNLM_PREC_F64) {
            dim3 grid(h->npad / ERR_TILE, jcnt);
            nb = (size_t)grid.x * grid.y;
            errors_kernel<double><<<grid, 256, 0, st>>>((const double *)h->A, h->npad, miss, h->W64, h->npad, h->H64, h->mpad, h->n, h->m, k4, h->partials, jt0);
        } else if (err_generic) {
            dim3 grid(h->npad / ERR_TILE, jcnt);
            nb = (size_t)grid.x * grid.y;
            errors_kernel<float><<<grid, 256, 0, st>>>((const float *)h->A, h->npad, miss, h->W64, h->npad, h->H64, h->mpad, h->n, h->m, k4, h->partials, jt0);
        } else {
            const int nx = h->npad / ERRF_TILE;
            const unsigned grid = 8u * ((nx + 7) / 8) * jcnt; // (XCD-aware numbering: k_errors.h)
            nb = (size_t)nx * jcnt;
            const int k2 = round_up_i(h->k, 2);
            const int lds = 2 * k2 * ERRF_TILE * (int)sizeof(float);
            const size_t cnt = (size_t)h->KP * h->mpad;
            factor_to_f32_kernel<<<(unsigned)((cnt + 255) / 256), 256, 0, st>>>(h->H64, cnt, h->Hkq);
            if (h->any_missing) {
                set_dyn_lds((const void *)errors_f32_kernel<true>, lds, "errors_f32_kernel");
                errors_f32_kernel<true><<<grid, 256, lds, st>>>((const float *)h->A, h->npad, h->missT, h->mpad / 32, (const float *)h->Wop, h->npad, h->Hkq,
                                                                h->mpad, h->n, h->m, k2, h->partials, jt0, nx);
            } else {
                set_dyn_lds((const void *)errors_f32_kernel<false>, lds, "errors_f32_kernel");
                errors_f32_kernel<false><<<grid, 256, lds, st>>>((const float *)h->A, h->npad, nullptr, 0, (const float *)h->Wop, h->npad, h->Hkq, h->mpad,
                                                                 h->n, h->m, k2, h->partials, jt0, nx);
            }
        }
        reduce_partials_kernel<<<1, REDUCE_THREADS, 0, st>>>(h->partials, nb, 2, h->scal);
        if (h->sharded && h->comm) {
            ncclResult_t r = g_rccl.AllReduce(h->scal, h->scal, 2, ncclDouble, ncclSum, (ncclComm_t)h->comm, st);
            if (r != ncclSuccess) return fail(h, NNLM_ERR_COMM, "ncclAllReduce (error sums) failed");
        }
    }
    if (need_pen) {
        const int nbw = (h->n + 255) / 256, nbh = (h->m + 255) / 256;
        penalty_kernel<<<nbw, 256, 0, st>>>(h->W64, h->npad, h->n, h->k, h->partials);
        reduce_partials_kernel<<<1, REDUCE_THREADS, 0, st>>>(h->partials, (size_t)nbw, 3, h->scal + 2);
        penalty_kernel<<<nbh, 256, 0, st>>>(h->H64, h->mpad, h->m, h->k, h->partials);
        reduce_partials_kernel<<<1, REDUCE_THREADS, 0, st>>>(h->partials, (size_t)nbh, 3, h->scal + 5);
    } else // (not recomputed: the collector must never see sums of an earlier call -- penalties() and need_pen could drift apart)
        HIPCHK(h, hipMemsetAsync(h->scal + 2, 0, 6 * sizeof(double), st));
    HIPCHK(h, hipMemcpyAsync(h->host_res, h->scal, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (with_sweeps) {
        const unsigned long long *src = h->sweeps + h->sw_active;
        if (h->sharded && h->comm) { // every rank swept its own columns: sum the integer counters
            HIPCHK(h, hipMemcpyAsync(h->sweeps_tmp, src, sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
            ncclResult_t r = g_rccl.AllReduce(h->sweeps_tmp, h->sweeps_tmp, 1, ncclUint64, ncclSum, (ncclComm_t)h->comm, st);
            if (r != ncclSuccess) return fail(h, NNLM_ERR_COMM, "ncclAllReduce (sweep counter) failed");
            src = h->sweeps_tmp;
        }
        HIPCHK(h, hipMemcpyAsync(h->host_res + 8, src, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipMemsetAsync(h->sweeps + h->sw_active, 0, sizeof(unsigned long long), st));
    }
    HIPCHK(h, hipEventRecord(h->ev_err, st));
    LAUNCHCHK(h);
    return NNLM_OK;
}

static int errors_collect(nnlm_handle *h, double *mse, double *mkl_var, double pen[6], long long *raw_sweeps)
{
    HIPCHK(h, hipEventSynchronize(h->ev_err));
    if (mse) *mse = h->host_res[0] / h->n_non_missing;
    if (mkl_var) *mkl_var = h->host_res[1] / h->n_non_missing;
    if (pen)
        for (int i = 0; i < 6; i++) pen[i] = h->host_res[2 + i];
    if (raw_sweeps) {
        unsigned long long v;
        memcpy(&v, h->host_res + 8, sizeof v);
        *raw_sweeps = (long long)v;
    }
    return NNLM_OK;
}

extern "C" int nnlm_errors(nnlm_handle *h, double *mse, double *mkl_var, double pen[6])
{
    if (!h || !h->A || !h->W64) return fail(h, NNLM_ERR_ARG, "nnlm_errors: matrix and factors must be set first");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = errors_launch(h, h->stream, false, 0, true);
    if (rc != NNLM_OK) return rc;
    return errors_collect(h, mse, mkl_var, pen, nullptr);
}

// ---------------------------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------------------------
extern "C" int nnlm_profile_enable(nnlm_handle *h, int on)
{
    if (!h) return fail(nullptr, NNLM_ERR_ARG, "handle is NULL");
    prof_collect(h);
    h->prof = on != 0;
    return NNLM_OK;
}
extern "C" int nnlm_profile_reset(nnlm_handle *h)
{
    if (!h) return fail(nullptr, NNLM_ERR_ARG, "handle is NULL");
    prof_collect(h);
    for (int i = 0; i < P_COUNT; i++) {
        h->prof_ms[i] = 0;
        h->prof_n[i] = 0;
    }
    return NNLM_OK;
}
extern "C" int nnlm_profile_get(nnlm_handle *h, const char *name, double *total_ms, long long *launches)
{
    if (!h || !name) return fail(h, NNLM_ERR_ARG, "nnlm_profile_get: bad arguments");
    prof_collect(h);
    for (int i = 0; i < P_COUNT; i++)
        if (strcmp(name, kProfNames[i]) == 0) {
            if (total_ms) *total_ms = h->prof_ms[i];
            if (launches) *launches = h->prof_n[i];
            return NNLM_OK;
        }
    return fail(h, NNLM_ERR_ARG, "nnlm_profile_get: unknown kernel class '%s'", name);
}

// ---------------------------------------------------------------------------------------------
// multi-GPU: RCCL, loaded lazily (single-GPU runs never touch librccl)
// ---------------------------------------------------------------------------------------------
extern "C" int nnlm_comm_unique_id(char id[NNLM_COMM_ID_BYTES])
{
    static_assert(NNLM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id) return fail(nullptr, NNLM_ERR_ARG, "nnlm_comm_unique_id: id is NULL");
    int rc = rccl_load();
    if (rc != NNLM_OK) return rc;
    ncclUniqueId u;
    ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return fail(nullptr, NNLM_ERR_COMM, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    memcpy(id, u.internal, NNLM_COMM_ID_BYTES);
    return NNLM_OK;
}

// id == NULL: "virtual rank" -- the handle computes rank's slab of an nranks-way split and leaves the partial
// [G | C] buffer un-reduced (single-device tests of the shard arithmetic, see nnlm_debug_partial).
extern "C" int nnlm_comm_init(nnlm_handle *h, const char id[NNLM_COMM_ID_BYTES], int rank, int nranks)
{
    if (!h) return fail(nullptr, NNLM_ERR_ARG, "nnlm_comm_init: handle is NULL");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(h, NNLM_ERR_ARG, "nnlm_comm_init: bad rank %d of %d", rank, nranks);
    HIPCHK(h, hipSetDevice(h->device));
    if (h->comm) {
        g_rccl.CommDestroy((ncclComm_t)h->comm);
        h->comm = nullptr;
    }
    h->rank = rank;
    h->nranks = nranks;
    invalidate_factor_caches(h); // (what an unpack left behind belongs to the previous communicator's exchange)
    h->pack_tail = 0;
    h->sharded = nranks > 1 || (id != nullptr); // a real 1-rank communicator runs the sharded path on one GPU (tests)
    h->dense_cols = true;                        // NNLM_FORM_COLS; nnlm_comm_set_form chooses the all-reduce form
    if (!id) return NNLM_OK;
    int rc = rccl_load();
    if (rc != NNLM_OK) return rc;
    ncclUniqueId u;
    memcpy(u.internal, id, NNLM_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, u, rank);
    if (r != ncclSuccess) return fail(h, NNLM_ERR_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    h->comm = comm;
    return NNLM_OK;
}

extern "C" int nnlm_comm_set_form(nnlm_handle *h, int form)
{
    if (!h) return fail(nullptr, NNLM_ERR_ARG, "nnlm_comm_set_form: handle is NULL");
    if (form != NNLM_FORM_COLS && form != NNLM_FORM_REDUCE) return fail(h, NNLM_ERR_ARG, "nnlm_comm_set_form: unknown form %d", form);
    if (h->dense_cols != (form == NNLM_FORM_COLS)) {
        HIPCHK(h, hipSetDevice(h->device));
        sync_all(h);
        invalidate_factor_caches(h); // (Gram, max and split copy an unpack left behind belong to the other form's exchange)
        h->pack_tail = 0;
        h->dense_cols = form == NNLM_FORM_COLS;
    }
    return NNLM_OK;
}

extern "C" int nnlm_get_info(nnlm_handle *h, const char *key, double *value)
{
    if (!h || !key || !value) return fail(h, NNLM_ERR_ARG, "nnlm_get_info: NULL argument");
    if (strcmp(key, "cus") == 0) *value = h->cus;
    else if (strcmp(key, "sweep_form_w") == 0) *value = h->sweep_form[0]; // (0 / 1 strict kernels, 2 k_sweep_f.h, 3 k_sweep_r.h)
    else if (strcmp(key, "sweep_form_h") == 0) *value = h->sweep_form[1];
    else if (strcmp(key, "sweep_groups_w") == 0) *value = h->sweep_groups[0];
    else if (strcmp(key, "sweep_groups_h") == 0) *value = h->sweep_groups[1];
    else if (strcmp(key, "kl_form_w") == 0) *value = h->kl_form[0];
    else if (strcmp(key, "kl_form_h") == 0) *value = h->kl_form[1];
    else return fail(h, NNLM_ERR_ARG, "nnlm_get_info: unknown key '%s'", key);
    return NNLM_OK;
}

extern "C" int nnlm_comm_info(nnlm_handle *h, int *rank, int *nranks)
{
    if (!h) return fail(nullptr, NNLM_ERR_ARG, "handle is NULL");
    if (rank) *rank = h->rank;
    if (nranks) *nranks = h->nranks;
    return NNLM_OK;
}

// Contraction range [begin, end) (in rows i of A for which = 1, columns j for which = 0) that rank `rank` of
// `nranks` owns.  Pure function of the sizes: the partition is stage-granular (a stage = 256 bytes of an A column for
// the TN kernel, 32 columns for the NT kernel), clipped to the true extent.
extern "C" int nnlm_shard_range(int n, int m, int precision, int which, int rank, int nranks, int *begin, int *end)
{
    if (n < 1 || m < 1 || nranks < 1 || rank < 0 || rank >= nranks || (which != 0 && which != 1) || !begin || !end)
        return fail(nullptr, NNLM_ERR_ARG, "nnlm_shard_range: bad arguments");
    nnlm_handle t;
    t.prec = precision;
    t.x16 = x16_enabled(precision);
    t.n = n;
    t.m = m;
    t.npad = round_up_i(n, NNLM_PAD_N);
    t.mpad = round_up_i(m, NNLM_PAD_M);
    const HalfPlan p = plan_half(&t, which, rank, nranks);
    const int CE = stage_elems(&t, which);
    const int lim = (which == 1) ? n : m;
    int c0 = p.stage_begin * CE, c1 = p.stage_end * CE;
    if (c1 > lim) c1 = lim;
    if (c0 > c1) c0 = c1;
    *begin = c0;
    *end = c1;
    return NNLM_OK;
}

// Columns [col0, col1) of the factor being solved that `rank` of `nranks` sweeps, and the slab width cpr of the all-gather
// (every rank sends [k][cpr]).  Pure function of the sizes.
extern "C" int nnlm_shard_cols(int ncols, int rank, int nranks, int *cpr, int *col0, int *col1)
{
    if (ncols < 1 || nranks < 1 || rank < 0 || rank >= nranks || !cpr || !col0 || !col1) return fail(nullptr, NNLM_ERR_ARG, "nnlm_shard_cols: bad arguments");
    nnlm_handle t;
    t.rank = rank;
    t.nranks = nranks;
    const ShardCols sc = shard_cols(&t, ncols);
    *cpr = sc.cpr;
    *col0 = sc.col0;
    *col1 = sc.col1;
    return NNLM_OK;
}

// ---------------------------------------------------------------------------------------------
// one-shot drivers
// ---------------------------------------------------------------------------------------------
// Arithmetic mode of the one-shot entries (= what the R layer reaches through r_glue.c): the reference is fp64 throughout
// and its own testthat vectors are held at 1.5e-8, so the drop-in default is the strict fp64 mode; the fp32-operand mode
// (A in 4 bytes per element, split-fp16 cross products; what bench.py times on a resident handle) is an explicit opt-in:
// NNLM_PRECISION=f32.
static int env_precision()
{
    const char *e = getenv("NNLM_PRECISION");
    if (e && (strcmp(e, "f32") == 0 || strcmp(e, "fp32") == 0 || strcmp(e, "0") == 0)) return NNLM_PREC_F32;
    return NNLM_PREC_F64;
}
static int env_device()
{
    const char *e = getenv("NNLM_DEVICE");
    return e ? atoi(e) : 0;
}

struct Lcg {
    uint64_t s = 0x9E3779B97F4A7C15ull;
    double next()
    {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(s >> 11) * (1.0 / 9007199254740992.0);
    }
};

static void cb_print(const nnlm_callbacks *cb, const char *fmt, ...)
{
    if (!cb || !cb->print) return;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    cb->print(cb->ctx, buf);
}

// add_penalty, src/nnmf.cpp:224-240 (same order of the six terms)
static double penalties(double terr, const double pen[6], double N, const double alpha[3], const double beta[3])
{
    if (alpha[0] != alpha[1]) terr += 0.5 * (alpha[0] - alpha[1]) * pen[0] / N;
    if (beta[0] != beta[1]) terr += 0.5 * (beta[0] - beta[1]) * pen[3] / N;
    if (alpha[1] != 0) terr += 0.5 * alpha[1] * pen[1] / N;
    if (beta[1] != 0) terr += 0.5 * beta[1] * pen[4] / N;
    if (alpha[2] != 0) terr += alpha[2] * pen[2] / N;
    if (beta[2] != 0) terr += beta[2] * pen[5] / N;
    return terr;
}

// The alternating loop of c_nnmf (reference src/nnmf.cpp:100-209) on a resident handle: matrix and factors are already in
// HBM.  Semantics are the reference's, iteration by iteration.  What is MI355X-specific is the schedule at a trace
// iteration: the error block runs on its own stream (HBM/MFMA bound) while the NEXT iteration's W half-step is enqueued
// speculatively on the main stream (its sweep is VALU/matrix-core bound and touches no HBM to speak of) into the second
// W buffer; when the host has the error block's result it either accepts the speculative half-step (swap buffers) or,
// if the stopping rule fired, drops it -- W_i is still intact in the first buffer.
extern "C" int nnlm_run(nnlm_handle *h, const double alpha[3], const double beta[3], unsigned max_iter, double rel_tol, int verbose,
                        int show_warning, unsigned inner_max_iter, double inner_rel_tol, int method, unsigned trace,
                        double *mse_error, double *mkl_error, double *target_error, double *average_epoch, int *n_trace,
                        unsigned *n_iteration, int *warned, const nnlm_callbacks *cb)
{
    if (!h || !h->A || !h->W64) return fail(h, NNLM_ERR_ARG, "nnlm_run: matrix and factors must be set first");
    if (!alpha || !beta || !mse_error || !mkl_error || !target_error || !average_epoch || !n_trace || !n_iteration || !warned)
        return fail(h, NNLM_ERR_ARG, "nnlm_run: NULL argument");
    if (method < 1 || method > 4) return fail(h, NNLM_ERR_ARG, "method must be 1..4 (got %d)", method);
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
#define CHK(x)                  \
    do {                        \
        rc = (x);               \
        if (rc != NNLM_OK) {    \
            g_last_error = h->err; \
            return rc;          \
        }                       \
    } while (0)

    if (trace < 1) trace = 1; // src/nnmf.cpp:53
    const unsigned err_len = nnlm_trace_capacity(max_iter, trace);
    const double N = h->n_non_missing;
    const int n = h->n, m = h->m;
    for (unsigned e = 0; e < err_len; e++) mkl_error[e] = h->kl_const; // src/nnmf.cpp:70,73

    if (verbose == 2) { // src/nnmf.cpp:100-104
        cb_print(cb, "\n%10s | %10s | %10s | %10s | %10s\n", "Iteration", "MSE", "MKL", "Target", "Rel. Err.");
        cb_print(cb, "--------------------------------------------------------------\n");
    }

    // the six penalty sums are read by penalties() only through these conditions (src/nnmf.cpp:224-240)
    const bool need_pen = alpha[0] != alpha[1] || beta[0] != beta[1] || alpha[1] != 0 || beta[1] != 0 || alpha[2] != 0 || beta[2] != 0;
    double rel_err = rel_tol + 1, terr_last = 1e99;
    unsigned i = 0, i_e = 0;
    auto book = [&](unsigned it, double mse, double kl, const double pen[6], long long raw) {
        mse_error[i_e] = mse;
        mkl_error[i_e] += kl;
        average_epoch[i_e] = (double)raw / (double)(n + m); // src/nnmf.cpp:145
        double t = (method < 3) ? 0.5 * mse_error[i_e] : mkl_error[i_e];
        t = penalties(t, pen, N, alpha, beta);
        target_error[i_e] = t;
        rel_err = 2 * (terr_last - t) / (terr_last + t + NNLM_TINY); // src/nnmf.cpp:153
        terr_last = t;
        if (verbose == 2) cb_print(cb, "%10d | %10.4f | %10.4f | %10.4f | %10.g\n", it + 1, mse_error[i_e], mkl_error[i_e], t, rel_err);
        ++i_e;
    };

    // spec.pending: the W half-step of iteration i is already enqueued (speculatively).  Whatever way this function is left
    // with one pending (stopping rule, interrupt, a failed call), the guard drops it: W_i is untouched, the alternate sweep
    // counter is cleared and the "what the last sweep left behind" state forgets the dropped sweep.
    struct SpecDrop {
        nnlm_handle *h;
        bool pending = false;
        void drop()
        {
            if (!pending) return;
            h->sg_which = h->sg_other = -1;
            h->gshard_for = h->upk_max_for = h->y16_for = -1; // (multi-GPU: what the dropped half-step's unpack left behind describes a W that is not current)
            h->fuse_err = false;
            h->fused_nb = 0;
            hipMemsetAsync(h->sweeps + (h->sw_active ^ 1), 0, sizeof(unsigned long long), h->stream);
            pending = false;
        }
        ~SpecDrop() { drop(); }
    } spec{h};
    for (; i < max_iter && std::fabs(rel_err) > rel_tol; i++) { // src/nnmf.cpp:109
        if (cb && cb->check_interrupt && cb->check_interrupt(cb->ctx)) { // src/nnmf.cpp:111
            sync_all(h);
            g_last_error = "interrupted";
            return NNLM_ERR_INTERRUPT;
        }
        if (verbose == 1 && cb && cb->progress) cb->progress(cb->ctx, i + 1, max_iter); // src/nnmf.cpp:112
        if (spec.pending) { // accept: its buffers and its sweep counter become the current ones
            swap_w(h);
            h->sw_active ^= 1;
            spec.pending = false;
        } else
            CHK(half_step(h, 0, alpha, inner_max_iter, inner_rel_tol, method)); // update W, src/nnmf.cpp:131
        CHK(half_step(h, 1, beta, inner_max_iter, inner_rel_tol, method));      // update H, src/nnmf.cpp:133
        if (i % trace == 0) {                                                   // src/nnmf.cpp:135-160
            HIPCHK(h, hipEventRecord(h->ev_hdone, h->stream));
            HIPCHK(h, hipStreamWaitEvent(h->stream_e, h->ev_hdone, 0));
            // (multi-GPU: the dense square-loss flows of the fp32-operand mode at rank <= 64 -- every rank's speculative cross product
            //  covers its own part of A, so the fused sums are a partition of the error sums; the accept / drop decision is a function
            //  of the all-reduced sums and therefore the same on every rank)
            const bool spec_sharded = h->sharded && h->x16 && !generic_rank(h) && !h->any_missing;
            h->err_launched = false;
            if (i + 1 < max_iter && (!h->sharded || spec_sharded) && method < 3) {
                // speculative W half-step of iteration i+1 (not made current: h->W64 still is W_i below).  The error block
                // (one more pass over A) starts together with it: since the cross product became HBM bound (k_xprod16.h) the
                // two streams of A share the bandwidth, but the error block is then finished before the latency-bound sweep
                // needs the CUs -- measured +2 % over holding it back until the cross product is done
                // Split-fp16 mode without missing values: that half-step's cross product streams A with H_i as its fixed
                // factor while W_i is still current, so it evaluates the error sums of (W_i, H_i) on the way
                // (xprod16_err_kernel) and no separate pass over A is needed.
                h->fuse_err = h->x16 && !generic_rank(h);
                h->fused_nb = 0;
                h->err_hook = h->sharded;
                h->err_need_pen = need_pen;
                rc = half_step(h, 0, alpha, inner_max_iter, inner_rel_tol, method, false, true);
                h->fuse_err = false;
                h->err_hook = false;
                if (rc != NNLM_OK) {
                    g_last_error = h->err;
                    return rc;
                }
                spec.pending = true;
                // Strict mode: errors64_kernel and the fp64 cross product both live on the fp64 matrix / vector pipe and serialise when they
                // share the chip (0.98 ms for the pair against 0.40 + 0.60).  The error block therefore starts when the speculative cross
                // product has finished streaming A -- beside the SWEEP, which is bound by dependent latency and leaves the pipe half idle.
                // (round 5, bench.py --precision f64: 1.506 against 1.516-1.526 ms per step; the cross product 0.40 instead of 0.69 ms, the error
                //  block 0.42 instead of 0.34)
                // (tried for the NA flow of the F32 mode too, scripts/gpu_r5_n.sh: the cross product 0.29 -> 0.16 ms, the Gram + solver kernels
                //  1.05 -> 1.21 beside errors_f32_kernel -- 2.277 against 2.268 ms per iteration: the work is conserved, not hidden)
                if (h->prec == NNLM_PREC_F64 && !h->fused_nb) HIPCHK(h, hipStreamWaitEvent(h->stream_e, h->ev_xdone, 0));
            } else
                h->fused_nb = 0;
            if (!h->err_launched) CHK(errors_launch(h, h->stream_e, true, h->fused_nb, need_pen)); // reads W_i, H_i and the active sweep counter (then zeroes it)
            h->err_launched = false;
            h->fused_nb = 0;
            // H (and the fp32 copy the error kernel reads) must not be rewritten before the error block is done
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_err, 0));
            double mse, kl, pen[6];
            long long raw = 0;
            CHK(errors_collect(h, &mse, &kl, pen, &raw));
            book(i, mse, kl, pen, raw);
        }
    }
    spec.drop(); // the stopping rule fired: the speculative half-step (if any) is discarded, W_i is untouched
    if ((unsigned)(i - 1) % trace != 0) { // src/nnmf.cpp:164 (unsigned arithmetic)
        double mse, kl, pen[6];
        long long raw = 0;
        CHK(errors_launch(h, h->stream, true, 0, need_pen)); // (error sums and the sweep counter in one round trip)
        CHK(errors_collect(h, &mse, &kl, pen, &raw));
        book(i, mse, kl, pen, raw);
    }

    if (verbose == 2) { // src/nnmf.cpp:194-198
        cb_print(cb, "--------------------------------------------------------------\n");
        cb_print(cb, "%10s | %10s | %10s | %10s | %10s\n\n", "Iteration", "MSE", "MKL", "Target", "Rel. Err.");
    }
    sync_all(h);
    LAUNCHCHK(h);
    *n_trace = (int)i_e;
    *n_iteration = i;
    *warned = (show_warning && rel_err > rel_tol) ? 1 : 0; // src/nnmf.cpp:208
    if (*warned && cb && cb->warning) cb->warning(cb->ctx, "Target tolerance not reached. Try a larger max.iter.");
    return NNLM_OK;
}

extern "C" int nnlm_c_nnmf(const double *A, int n, int m, unsigned k, const double *W_init, const double *H_init, const int *Wm,
                           const int *Hm, const double alpha[3], const double beta[3], unsigned max_iter, double rel_tol,
                           int n_threads, int verbose, int show_warning, unsigned inner_max_iter, double inner_rel_tol,
                           int method, unsigned trace, double *W_out, double *H_out, double *mse_error, double *mkl_error,
                           double *target_error, double *average_epoch, int *n_trace, unsigned *n_iteration, int *warned,
                           const nnlm_callbacks *cb)
{
    (void)n_threads;
    if (!A || !alpha || !beta || !W_out || !H_out || !mse_error || !mkl_error || !target_error || !average_epoch || !n_trace ||
        !n_iteration || !warned)
        return fail(nullptr, NNLM_ERR_ARG, "nnlm_c_nnmf: NULL argument");
    if (k < 1) return fail(nullptr, NNLM_ERR_ARG, "nnlm_c_nnmf: k must be >= 1");
    nnlm_handle *h = nullptr;
    int rc = nnlm_create(&h, env_device(), env_precision());
    if (rc != NNLM_OK) return rc;
    struct Guard {
        nnlm_handle *h;
        ~Guard() { nnlm_destroy(h); }
    } guard{h};
    CHK(nnlm_set_matrix(h, A, n, m));

    // default init, src/nnmf.cpp:82-98: W.randu(k,n)*0.01 drawn column-major, W first, masked entries zeroed
    std::vector<double> Wi, Hi;
    Lcg lcg;
    auto draw = [&]() { return (cb && cb->unif_rand) ? cb->unif_rand(cb->ctx) : lcg.next(); };
    if (!W_init) {
        Wi.resize((size_t)n * k);
        for (int i = 0; i < n; i++)
            for (unsigned q = 0; q < k; q++) {
                double v = draw() * 0.01;
                if (Wm && Wm[(size_t)q * n + i] != 0) v = 0.0; // (Wm > 0 on the reference's unsigned matrix: any non-zero, NA_LOGICAL included)
                Wi[(size_t)q * n + i] = v;
            }
        W_init = Wi.data();
    }
    if (!H_init) {
        Hi.resize((size_t)k * m);
        for (size_t e = 0; e < (size_t)k * m; e++) {
            double v = draw() * 0.01;
            if (Hm && Hm[e] != 0) v = 0.0;
            Hi[e] = v;
        }
        H_init = Hi.data();
    }
    CHK(nnlm_set_factors(h, k, W_init, H_init, Wm, Hm));
    CHK(nnlm_run(h, alpha, beta, max_iter, rel_tol, verbose, show_warning, inner_max_iter, inner_rel_tol, method, trace, mse_error,
                 mkl_error, target_error, average_epoch, n_trace, n_iteration, warned, cb));
    CHK(nnlm_get_factors(h, W_out, H_out));
    return NNLM_OK;
}

extern "C" int nnlm_c_nnlm(const double *x, const double *y, int n, int p, int q, const double alpha[3], const int *mask,
                           const double *beta0, unsigned max_iter, double rel_tol, int n_threads, int method, double *coefficient,
                           int *n_iteration, const nnlm_callbacks *cb)
{
    (void)n_threads;
    if (!x || !y || !alpha || !coefficient || !n_iteration) return fail(nullptr, NNLM_ERR_ARG, "nnlm_c_nnlm: NULL argument");
    if (n < 1 || p < 1 || q < 1) return fail(nullptr, NNLM_ERR_ARG, "nnlm_c_nnlm: empty x or y");
    nnlm_handle *h = nullptr;
    int rc = nnlm_create(&h, env_device(), env_precision());
    if (rc != NNLM_OK) return rc;
    struct Guard {
        nnlm_handle *h;
        ~Guard() { nnlm_destroy(h); }
    } guard{h};
    // y plays A (n x q), x plays W (n x p), beta plays H (p x q): one H half-step, src/nnlm.cpp:44-47
    CHK(nnlm_set_matrix(h, y, n, q));
    std::vector<double> b0;
    if (!beta0) { // beta.randu(), src/nnlm.cpp:38-39
        b0.resize((size_t)p * q);
        Lcg lcg;
        for (auto &v : b0) v = (cb && cb->unif_rand) ? cb->unif_rand(cb->ctx) : lcg.next();
        beta0 = b0.data();
    }
    CHK(nnlm_set_factors(h, (unsigned)p, x, beta0, nullptr, mask));
    CHK(half_step(h, 1, alpha, max_iter, rel_tol, method));
    long long raw = 0;
    CHK(nnlm_take_sweeps(h, &raw, 1));
    *n_iteration = (int)raw;
    CHK(nnlm_get_factors(h, nullptr, coefficient));
    return NNLM_OK;
#undef CHK
}
