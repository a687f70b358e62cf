// C-ABI of libgdca.so (see include/gdca.h): context, workspace, the fused device pipeline and
// the operator-level entry points.  Host-side orchestration only; all arithmetic is in the
// k_*.hip kernels.  No CPU fallback exists: every entry point needs a HIP device.
#include <algorithm>
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include "gdca_internal.h"
#include "gdca_launch.h"

#define MAX_EV 1024
#define N_SCRATCH 10

struct gdca_buf {
    void *p;
    size_t cap;
};

// Shared by the contexts of one pipeline (gdca_ctx_create_peer): orders their SPD-inverse stages one after
// the other on the device, so that only the non-MFMA stages of the next family overlap the inverse of the
// current one (two interleaved inverses would just halve each other's MFMA rate).
struct gdca_gate {
    hipEvent_t ev[4];
    int next, last, armed, refs;
    // the pipeline's own stream: every phase batch issued as batched grids by a member of the pipeline goes here, so that the batches
    // of two alternating context sets are ONE in-order sequence -- two streams that share a hardware queue take turns with hundreds
    // of microseconds between them (400 us measured between one set's score stage and the other's front end at config B), and on
    // different hardware queues the next batch's small kernels become resident beside the running persistent sweep
    hipStream_t batch_stream;
};

struct gdca_ctx {
    int device;
    gdca_gate *gate;
    hipStream_t stream;
    bool own_stream;
    bool timing;
    gdca_tuning tune;          // this context's switches (gdca_ctx_set_option; GDCA_* environment at creation)
    char err[512];
    // named device buffers (grow-only)
    gdca_buf Zt, Zp, hist, Zb, hcnt, nk, W, Wfix, Pifix, Pipc, A, G, H, P, Sg, Dblk, Ld, Tws, colsum, sc;
    gdca_buf normws, C2, B0, Rt;  // ||X||_1 workspace; Newton-Schulz refinement (allocated when a run first needs it): C again, X0 in full, I - X0 C
    gdca_buf Wd;                  // Cholesky fallback: the inverses of the diagonal tiles of the factor
    gdca_buf hcand;               // reweighting, bound forms: the list of candidate pairs
    gdca_buf himg;                // reweighting, fp4 form: the image of the three low bit planes as E2M1 nibbles
    gdca_buf rankws;              // device ranking: keys, values, histograms, the three output arrays
    // an enqueued ranked run (gdca_run_ranked_async): where its ranking will be, and whether enqueueing it worked
    bool rank_pending = false;
    long long rank_len = 0;
    int rank_sep = 0;
    gdca_status rank_status = GDCA_OK;
    int32_t *rank_i = nullptr, *rank_j = nullptr;
    double *rank_s = nullptr;
    int ncu;                   // compute units of the device
    int *item0_host;           // pinned staging of the sweep's item table
    int item0_cap;
    gdca_buf scratch[N_SCRATCH];
    gdca_dev_scalars *sc_host;  // pinned
    gdca_dev_scalars *sc_host_dev;  // ... as the device addresses it (k_publish_scalars)
    bool sc_published;          // the enqueued run ends with k_publish_scalars: its collect reads sc_host after a stream synchronisation
    hipEvent_t ev[MAX_EV];
    int n_ev;
    // state of an enqueued, not yet collected run (gdca_run_dev_async / gdca_run_collect)
    hipEvent_t ev_batch, ev_upload;
    bool pending;
    bool pend_timed;
    bool stamped;              // the run being enqueued marks its stages with device time stamps (sc->stamp) instead of HIP events: a member of a batch issued as batched grids
    bool pend_inv_stamped;     // the enqueued run's inverse was a merged launch bracketed by stamps (slots 6 / 16 before, 17 / 4 behind) instead of events
    bool pend_stamped;         // ... and the enqueued run did so: its collect reads the stamps
    int pend_score_batch;      // ... and of its score grids
    int pend_front_batch;      // members that shared this run's batched front-end / score grids (1: grids of its own); their stage times are the grids' divided by it
    bool pend_fn_timed, pend_tally_timed;  // events 7 / 8 around k_fn, 9 / 10 around k_pair_tally were recorded by this run
    int pend_N, pend_M, pend_q, pend_n, pend_npad, pend_nupd;
    int pend_batch;            // families that shared this run's SPD-inverse launch (1: a launch of its own)
    const int8_t *pend_Z;      // what a refinement at collect time needs to build C again and to score again
    double *pend_S;
    gdca_params pend_p;
    int pend_refined;
    int pend_attempt;          // attempts the enqueued run's inverse has had so far (0: the first is enqueued); its watchdog may end a launch (k_inverse.hip, spin_until)
    bool pend_rescored;        // ... and it was run again at collect time: the scores (and a ranking) are those of the last attempt
    hipEvent_t pend_upd_ev[2]; // the two events around that launch (this context's own, also for a merged launch)
    double pend_upd_flops;
};

// ---- batched grids: the recorder of gdca_launch.h, and the two stream operations of this file in recordable form ------------------
gdca_recorder *&gdca_recorder::active()
{
    static thread_local gdca_recorder *r = nullptr;
    return r;
}

void gdca_recorder::begin(hipStream_t s, int members)
{
    stream = s;
    lists.assign((size_t)members, {});
    cur = -1;
    active() = this;
}

void gdca_recorder::add(const gdca_op &op)
{
    lists[(size_t)cur].push_back(op);
    ++ops;
}

hipError_t gdca_recorder::flush()
{
    const int K = (int)lists.size();
    std::vector<size_t> at((size_t)K, 0);
    hipError_t err = hipSuccess;
    const int keep = cur;
    cur = -1;  // (the launchers below go out at once)
    for (;;) {
        int k0 = -1;
        for (int k = 0; k < K; ++k)
            if (at[k] < lists[k].size() && (k0 < 0 || at[k] < at[k0])) k0 = k;
        if (k0 < 0) break;
        // the kind at the head of the list of the member that has got least far, and every member whose head is of that kind
        const gdca_op &h = lists[k0][at[k0]];
        const gdca_op *grp[GDCA_MAXB];
        int n = 0;
        for (int k = 0; k < K && n < GDCA_MAXB; ++k) {
            if (at[k] >= lists[k].size()) continue;
            const gdca_op &o = lists[k][at[k]];
            if (o.launch != h.launch || o.block.x != h.block.x || o.block.y != h.block.y || o.block.z != h.block.z) continue;
            grp[n++] = &o;
            ++at[k];
        }
        h.launch(stream, grp, n);
        ++launches;
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess && err == hipSuccess) err = e;
    }
    for (auto &l : lists) l.clear();
    cur = keep;
    return err;
}

hipError_t gdca_recorder::end()
{
    const hipError_t e = flush();
    active() = nullptr;
    cur = -1;
    return e;
}

struct k_fill32_args {
    unsigned *p;
    unsigned v;
    size_t words;
};
template <int CAP>
__global__ __launch_bounds__(256) void k_fill32(const BatchArgs<k_fill32_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    unsigned *__restrict__ p = a_.p;
    const unsigned v = a_.v;
    const size_t words = a_.words;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = v;
}

// hipMemsetAsync for buffers of whole 32-bit words; while the calling thread records a batch, a kernel of the batch
void gdca_fill_async(hipStream_t s, void *p, int byte_value, size_t bytes)
{
    gdca_recorder *r = gdca_recorder::active();
    if (!r || r->cur < 0 || s != r->stream || (bytes & 3) || ((uintptr_t)p & 3)) {
        GDCA_FLUSH_RECORDED();
        (void)hipMemsetAsync(p, byte_value, bytes, s);
        return;
    }
    const size_t words = bytes / 4;
    const unsigned v = (unsigned)(byte_value & 0xff) * 0x01010101u;
    const unsigned grid = (unsigned)std::min<size_t>((words + 255) / 256, 1024);
    if (grid) gdca_launch<k_fill32_args, k_fill32<1>, k_fill32<GDCA_MAXB>>(dim3(grid), dim3(256), 0, s, k_fill32_args{(unsigned *)p, v, words});
}

struct k_stamp_args {
    gdca_dev_scalars *sc;
    int slot, slot2;
};
template <int CAP>
__global__ __launch_bounds__(64) void k_stamp(const BatchArgs<k_stamp_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    if (threadIdx.x == 0) {
        const unsigned long long t = wall_clock64();
        a_.sc->stamp[a_.slot] = t;
        if (a_.slot2 >= 0) a_.sc->stamp[a_.slot2] = t;
    }
}

void gdca_launch_stamp(hipStream_t s, gdca_dev_scalars *sc, int slot, int slot2)
{
    gdca_launch<k_stamp_args, k_stamp<1>, k_stamp<GDCA_MAXB>>(dim3(1), dim3(64), 0, s, k_stamp_args{sc, slot, slot2});
}

// ---- tuning switches: environment at context creation, gdca_ctx_set_option afterwards -----------------------------------------
static bool parse_long(const char *v, long *out)
{
    if (!v || !*v) return false;
    char *end = nullptr;
    const long x = strtol(v, &end, 10);
    if (end == v || *end) return false;
    *out = x;
    return true;
}

bool gdca_tuning_set(gdca_tuning *t, const char *key, const char *value)
{
    if (!t || !key || !value) return false;
    char k[64];
    size_t n = 0;
    if (!strncasecmp(key, "GDCA_", 5)) key += 5;
    for (; key[n] && n + 1 < sizeof(k); ++n) k[n] = (char)toupper((unsigned char)key[n]);
    k[n] = 0;
    if (!strcmp(k, "SWEEP_TRACE")) {
        if (strlen(value) >= sizeof(t->sweep_trace)) return false;
        strcpy(t->sweep_trace, value);
        return true;
    }
    if (!strcmp(k, "HAMMING_MODE")) {
        const char c = (char)tolower((unsigned char)value[0]);
        t->hamming_mode = c == 'f' ? 0 : (c == 'b' ? 1 : (c == 'm' ? 2 : -1));
        return c == 'f' || c == 'b' || c == 'm' || c == 'a' || c == 0;  // full | bound | mfma | auto
    }
    if (!strcmp(k, "FORCE_FALLBACK")) {  // any value but "" and "0" switches it on (as DCAUTILS_FORCE_FALLBACK in the reference's tests)
        t->force_fallback = (*value && strcmp(value, "0") != 0) ? 1 : 0;
        return true;
    }
    long x = 0;
    if (strcmp(k, "REFINE") && strcmp(k, "REFINE_COND") && !parse_long(value, &x)) return false;
    struct { const char *name; int *field; long lo, hi; } ints[] = {
        {"GROUP", &t->group, -1, 4},        {"RAMP", &t->ramp, 0, 1},          {"RAGGED", &t->ragged, 0, 1},
        {"REM_TAIL", &t->rem_tail, -1, 1 << 20}, {"PANEL_HALVES", &t->panel_halves, -1, 1}, {"SLAB", &t->slab, 0, 1},
        {"RING", &t->ring, 2, 8},           {"MCUS", &t->mcus, -1, 32},        {"MCU_SOLO", &t->mcu_solo, -1, 1},       {"SWEEP_DEBUG", &t->sweep_debug, 0, 63}, {"SWEEP_RETRIES", &t->sweep_retries, 0, 5},
        {"TALLY_TJ", &t->tally_tj, 0, 32},  {"MERGE", &t->merge, 1, 8},        {"MERGE_BLOCKS", &t->merge_blocks, 1, 64},
        {"MERGE_MCUS", &t->merge_mcus, -1, 16},  {"MERGE_GROUP", &t->merge_group, -1, 4}, {"MERGE_TILES", &t->merge_tiles, 1, 1 << 20},
        {"CHOLESKY", &t->cholesky, 0, 2},  {"PHASED_FRONTS", &t->phased_fronts, 0, 1}, {"PHASED_GRIDS", &t->phased_grids, -1, 8}, {"PHASED_STREAMS", &t->phased_streams, 1, 64},
    };
    for (auto &e : ints)
        if (!strcmp(k, e.name)) {
            if (x < e.lo || x > e.hi) return false;
            *e.field = (int)x;
            return true;
        }
    if (!strcmp(k, "REFINE")) {
        const char c = (char)tolower((unsigned char)value[0]);
        if (c == 'a' || !strcmp(value, "-1")) t->refine = -1;
        else if (!strcmp(value, "0") || !strcasecmp(value, "off") || c == 'n') t->refine = 0;
        else if (!strcmp(value, "1") || !strcasecmp(value, "on") || c == 'y') t->refine = 1;
        else return false;
        return true;
    }
    if (!strcmp(k, "REFINE_COND")) {
        char *end = nullptr;
        const double x = strtod(value, &end);
        if (end == value || *end || !(x > 0.0)) return false;
        t->refine_cond = x;
        return true;
    }
    if (!strcmp(k, "SWEEP_TIMEOUT_MS")) {
        if (x < 0) return false;
        t->sweep_timeout_ms = x;
        return true;
    }
    return false;
}

void gdca_tuning_from_env(gdca_tuning *t)
{
    memset(t, 0, sizeof(*t));
    t->group = -1;
    t->ramp = 1;
    t->ragged = 1;
    t->rem_tail = -1;
    t->panel_halves = -1;
    t->slab = 1;
    t->ring = 8;
    t->mcus = -1;
    t->mcu_solo = -1;
    t->sweep_retries = 2;
    t->hamming_mode = -1;
    t->merge = 8;
    t->merge_blocks = 57;
    t->merge_mcus = -1;
    t->merge_group = -1;
    t->merge_tiles = 2300;
    t->phased_fronts = 1;
    t->phased_grids = -1;
    t->phased_streams = 4;
    t->refine = -1;
    t->refine_cond = 1e6;
    t->cholesky = 1;
    static const char *const names[] = {"GDCA_GROUP", "GDCA_RAMP", "GDCA_RAGGED", "GDCA_REM_TAIL", "GDCA_PANEL_HALVES", "GDCA_SLAB",
                                        "GDCA_RING", "GDCA_MCUS", "GDCA_SWEEP_DEBUG", "GDCA_SWEEP_TIMEOUT_MS", "GDCA_SWEEP_RETRIES", "GDCA_TALLY_TJ",
                                        "GDCA_HAMMING_MODE", "GDCA_FORCE_FALLBACK", "GDCA_MERGE", "GDCA_MERGE_BLOCKS",
                                        "GDCA_MERGE_MCUS", "GDCA_MERGE_GROUP", "GDCA_MERGE_TILES", "GDCA_REFINE", "GDCA_REFINE_COND", "GDCA_CHOLESKY", "GDCA_SWEEP_TRACE", "GDCA_PHASED_FRONTS", "GDCA_PHASED_GRIDS", "GDCA_PHASED_STREAMS", "GDCA_MCU_SOLO"};
    for (const char *nm : names)
        if (const char *v = getenv(nm)) (void)gdca_tuning_set(t, nm, v);  // an unusable value leaves the default
}

static gdca_status fail(gdca_ctx *ctx, gdca_status st, const char *fmt, const char *a, const char *b)
{
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), fmt, a, b);
    return st;
}

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return fail(ctx, GDCA_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define CHK(expr)                         \
    do {                                  \
        gdca_status s_ = (expr);          \
        if (s_ != GDCA_OK) return s_;     \
    } while (0)

static gdca_status ensure(gdca_ctx *ctx, gdca_buf &b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return GDCA_OK;
    if (b.p) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(ctx, GDCA_ENOMEM, "hipMalloc failed: %s%s", hipGetErrorString(e), "");
    }
    b.cap = want;
    return GDCA_OK;
}

static gdca_status check_launch(gdca_ctx *ctx, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, GDCA_EHIP, "launch %s: %s", what, hipGetErrorString(e));
    return GDCA_OK;
}

extern "C" {

int32_t gdca_version(void)
{
    return GDCA_VERSION_MAJOR * 1000 + GDCA_VERSION_MINOR;
}

int32_t gdca_stats_bytes(void)
{
    return (int32_t)sizeof(gdca_stats);
}

int32_t gdca_params_bytes(void)
{
    return (int32_t)sizeof(gdca_params);
}

int32_t gdca_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

gdca_status gdca_ctx_create_on_stream(int32_t device_id, void *hip_stream, gdca_ctx **out)
{
    if (!out) return GDCA_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GDCA_EHIP;
    if (device_id < 0 || device_id >= ndev) return GDCA_EINVAL;
    gdca_ctx *ctx = (gdca_ctx *)calloc(1, sizeof(gdca_ctx));
    if (!ctx) return GDCA_ENOMEM;
    ctx->device = device_id;
    ctx->timing = true;
    gdca_tuning_from_env(&ctx->tune);
    if (hipSetDevice(device_id) != hipSuccess) {
        free(ctx);
        return GDCA_EHIP;
    }
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    {
        hipDeviceProp_t prop0;
        ctx->ncu = hipGetDeviceProperties(&prop0, device_id) == hipSuccess ? prop0.multiProcessorCount : 256;
    }
    // from here on every failure goes through gdca_ctx_destroy, which frees whatever has been created so far
    if (hipEventCreateWithFlags(&ctx->ev_batch, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming) != hipSuccess) {
        gdca_ctx_destroy(ctx);
        return GDCA_EHIP;
    }
    if (hipHostMalloc((void **)&ctx->sc_host, sizeof(gdca_dev_scalars), hipHostMallocDefault) != hipSuccess) {
        ctx->sc_host = nullptr;
        gdca_ctx_destroy(ctx);
        return GDCA_ENOMEM;
    }
    if (hipHostGetDevicePointer((void **)&ctx->sc_host_dev, ctx->sc_host, 0) != hipSuccess) ctx->sc_host_dev = nullptr;  // (then collect copies)
    *out = ctx;
    return GDCA_OK;
}

// A stream's hardware queue comes into being with the stream's FIRST command, not with the stream -- and the driver maps a new queue
// by taking every queue of the device off the hardware and putting them back: the waves of all running kernels are saved and restored
// (a standstill of ~1.7 ms, measured from inside the sweep kernel; the waves come back on other compute units).  Restored beside the
// kernels of other streams, a persistent sweep need not get all its workgroups back at once -- and those it did get back wait for
// the items the others hold, keeping the compute units those would need: the launch stands still until its watchdog ends it.
// Round 6's batch driver ran into exactly that, in 1 to 8 of 100 runs: the contexts of its phase batches were made when the first
// small family showed up, their streams started beside a slot's running sweep (tools/rounds/r06/gpu_r6q.sh, gpu_r6s.sh; made before
// any work: 0 of 200).  So the queue is made HERE, with the context -- one small fill and a synchronisation; the context's scalars
// are allocated on the way -- and callers make their contexts before they enqueue work (include/gdca.h).  What a caller cannot
// rule out (another process starting on the same GPU) is what the sweep's watchdog and gdca_run_collect's second attempt are for.
static gdca_status warm_stream(gdca_ctx *ctx, hipStream_t s)
{
    CHK(ensure(ctx, ctx->sc, sizeof(gdca_dev_scalars)));
    gdca_fill_async(s, ctx->sc.p, 0, sizeof(gdca_dev_scalars));
    if (hipStreamSynchronize(s) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, GDCA_EHIP, "the context's stream could not be started%s%s", "", "");
    }
    return GDCA_OK;
}

gdca_status gdca_ctx_create(int32_t device_id, gdca_ctx **out)
{
    gdca_status st = gdca_ctx_create_on_stream(device_id, nullptr, out);
    if (st != GDCA_OK) return st;
    gdca_ctx *ctx = *out;
    // the context's own stream: non-blocking, at the highest priority the device offers (a family's kernels should not queue
    // behind whatever else the process runs on the device)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_greatest) != hipSuccess) {
        gdca_ctx_destroy(ctx);
        *out = nullptr;
        return GDCA_EHIP;
    }
    ctx->own_stream = true;
    st = warm_stream(ctx, ctx->stream);
    if (st != GDCA_OK) {
        gdca_ctx_destroy(ctx);
        *out = nullptr;
    }
    return st;
}

gdca_status gdca_ctx_create_peer(gdca_ctx *leader, gdca_ctx **out)
{
    if (!leader || !out) return GDCA_EINVAL;
    gdca_ctx *ctx = leader;
    HIPCHK(hipSetDevice(leader->device));
    if (!leader->gate) {
        gdca_gate *g = (gdca_gate *)calloc(1, sizeof(gdca_gate));
        if (!g) return GDCA_ENOMEM;
        for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming));
        g->refs = 1;
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (hipStreamCreateWithPriority(&g->batch_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess) g->batch_stream = nullptr;
        leader->gate = g;
    }
    const bool new_gate = leader->gate->refs == 1;
    gdca_status st = gdca_ctx_create(leader->device, out);
    if (st != GDCA_OK) return st;
    (*out)->gate = leader->gate;
    leader->gate->refs += 1;
    if (new_gate && leader->gate->batch_stream) {  // (the pipeline's own stream too)
        st = warm_stream(*out, leader->gate->batch_stream);
        if (st != GDCA_OK) {
            gdca_ctx_destroy(*out);
            *out = nullptr;
        }
    }
    return st;
}

gdca_status gdca_ctx_destroy(gdca_ctx *ctx)
{
    if (!ctx) return GDCA_EINVAL;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream || ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    gdca_buf *bufs[] = {&ctx->Zt, &ctx->Zp, &ctx->hist, &ctx->Zb, &ctx->hcnt, &ctx->nk, &ctx->W, &ctx->Wfix, &ctx->Pifix,
                        &ctx->Pipc, &ctx->A, &ctx->G, &ctx->H, &ctx->P, &ctx->Sg, &ctx->Dblk, &ctx->Ld,
                        &ctx->Tws, &ctx->colsum, &ctx->sc, &ctx->normws, &ctx->C2, &ctx->B0, &ctx->Rt, &ctx->Wd, &ctx->rankws, &ctx->hcand, &ctx->himg};
    for (gdca_buf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < N_SCRATCH; ++i)
        if (ctx->scratch[i].p) (void)hipFree(ctx->scratch[i].p);
    if (ctx->gate && --ctx->gate->refs == 0) {
        for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ctx->gate->ev[i]);
        if (ctx->gate->batch_stream) {
            (void)hipStreamSynchronize(ctx->gate->batch_stream);
            (void)hipStreamDestroy(ctx->gate->batch_stream);
        }
        free(ctx->gate);
    }
    for (int i = 0; i < ctx->n_ev; ++i) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->item0_host) (void)hipHostFree(ctx->item0_host);
    if (ctx->ev_batch) (void)hipEventDestroy(ctx->ev_batch);
    if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
    if (ctx->sc_host) (void)hipHostFree(ctx->sc_host);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    free(ctx);
    return GDCA_OK;
}

gdca_status gdca_ctx_synchronize(gdca_ctx *ctx)
{
    if (!ctx) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GDCA_OK;
}

const char *gdca_last_error(gdca_ctx *ctx)
{
    return ctx ? ctx->err : "null context";
}

gdca_status gdca_ctx_set_timing(gdca_ctx *ctx, int32_t enabled)
{
    if (!ctx) return GDCA_EINVAL;
    ctx->timing = enabled != 0;
    return GDCA_OK;
}

gdca_status gdca_ctx_set_option(gdca_ctx *ctx, const char *key, const char *value)
{
    if (!ctx) return GDCA_EINVAL;
    if (!key || !value) return fail(ctx, GDCA_EINVAL, "null option key or value%s%s", "", "");
    if (!gdca_tuning_set(&ctx->tune, key, value)) return fail(ctx, GDCA_EINVAL, "unknown option or unusable value: %s=%s", key, value);
    return GDCA_OK;
}

}  // extern "C"

static gdca_status need_events(gdca_ctx *ctx, int n)
{
    if (n > MAX_EV) n = MAX_EV;
    while (ctx->n_ev < n) {
        HIPCHK(hipEventCreate(&ctx->ev[ctx->n_ev]));
        ++ctx->n_ev;
    }
    return GDCA_OK;
}

// A stage boundary of the run being enqueued: a HIP event of the context -- or, for a member of a batch whose kernels go out as
// batched grids, a device time stamp written by a kernel of the batch (slot = the event's index: one launch stamps all members)
static gdca_status mark(gdca_ctx *ctx, int slot)
{
    if (ctx->stamped) {
        gdca_launch_stamp(ctx->stream, (gdca_dev_scalars *)ctx->sc.p, slot, -1);
        return GDCA_OK;
    }
    HIPCHK(hipEventRecord(ctx->ev[slot], ctx->stream));
    return GDCA_OK;
}

// a member's failure as the LEADER's last error (the caller of a batch entry reads the leader's)
static gdca_status member_error(gdca_ctx *lead, gdca_ctx *m, int k, gdca_status st)
{
    if (k > 0 && m != lead && m->err[0]) {
        char msg[sizeof(lead->err)];
        snprintf(msg, sizeof(msg), "member %d: %.400s", k, m->err);
        memcpy(lead->err, msg, sizeof(msg));
    }
    return st;
}

// The host buffer of an upload just enqueued on ctx->stream may be reused by the caller: true at once for pageable memory (the copy
// is staged before hipMemcpyAsync returns), a DMA still in flight for pinned or registered memory -- wait for it (an event right
// behind the copy; the stream of a context that is not pending holds nothing else)
static gdca_status upload_done(gdca_ctx *ctx)
{
    HIPCHK(hipEventRecord(ctx->ev_upload, ctx->stream));
    HIPCHK(hipEventSynchronize(ctx->ev_upload));
    return GDCA_OK;
}

static double inverse_flops_model(double n)
{
    return (n * n * n / 3.0 + n * n / 2.0 + n / 6.0) + (2.0 * n * n * n / 3.0 + n * n / 2.0 + 5.0 * n / 6.0);
}

static int round_up(int x, int m)
{
    return (x + m - 1) / m * m;
}

// stage 1+2: theta, threshold, neighbour counts, W, Wfix, Meff  (all on device)
static gdca_status weights_stage(gdca_ctx *ctx, const int8_t *Zd, int N, int M, int q, double theta_in, int fixed_thresh,
                                 bool want_theta_only, int mark_after_theta)
{
    hipStream_t s = ctx->stream;
    gdca_dev_scalars *sc = (gdca_dev_scalars *)ctx->sc.p;
    CHK(ensure(ctx, ctx->hist, (size_t)N * 32 * sizeof(uint32_t)));
    if (fixed_thresh >= 0) {
        gdca_launch_set_thresh(s, sc, fixed_thresh);
    } else {
        if (theta_in < 0.0) {
            gdca_fill_async(s, ctx->hist.p, 0, (size_t)N * 32 * sizeof(uint32_t));
            gdca_launch_column_hist(s, Zd, (uint32_t *)ctx->hist.p, N, M);
        }
        gdca_launch_theta_finalize(s, (const uint32_t *)ctx->hist.p, N, M, theta_in, sc);
    }
    CHK(check_launch(ctx, "theta"));
    if (mark_after_theta >= 0) CHK(mark(ctx, mark_after_theta));
    if (want_theta_only) return GDCA_OK;

    const int Mt = (M + GDCA_HTILE - 1) / GDCA_HTILE;
    CHK(ensure(ctx, ctx->Zb, gdca_bitplane_bytes(N, M)));
    CHK(ensure(ctx, ctx->hcnt, (size_t)Mt * GDCA_HTILE * sizeof(int32_t)));
    if (!ctx->tune.force_fallback && ctx->tune.hamming_mode != 0) CHK(ensure(ctx, ctx->hcand, gdca_hamming_cand_cap(M) * 8));
    // (the fp4 form's image of the three low bit planes: 1.5 N bytes per sequence)
    // (only where the option asks for that form: the automatic choice never takes it -- k_hamming_fp4.hip says why)
    const bool fp4 = !ctx->tune.force_fallback && ctx->tune.hamming_mode == 2;
    if (fp4) CHK(ensure(ctx, ctx->himg, gdca_fp4_image_bytes(N, M)));
    CHK(ensure(ctx, ctx->nk, (size_t)M * sizeof(int32_t)));
    CHK(ensure(ctx, ctx->W, (size_t)M * sizeof(double)));
    CHK(ensure(ctx, ctx->Wfix, (size_t)M * sizeof(unsigned long long)));
    gdca_fill_async(s, ctx->hcnt.p, 0, (size_t)Mt * GDCA_HTILE * sizeof(int32_t));
    gdca_launch_bitplane_pack(s, Zd, (uint32_t *)ctx->Zb.p, N, M, q, sc);  // (also the symbol-range check)
    // option GDCA_FORCE_FALLBACK: the independent byte-compare kernel instead of the bit-sliced one -- the analogue of
    // DCAUTILS_FORCE_FALLBACK in the reference's tests (test/runtests.jl:78-86)
    if (ctx->tune.force_fallback)
        gdca_launch_hamming_fallback(s, Zd, (int32_t *)ctx->hcnt.p, N, M, sc);
    else
        gdca_launch_hamming(s, (const uint32_t *)ctx->Zb.p, Zd, (int32_t *)ctx->hcnt.p, N, M, sc, ctx->tune.hamming_mode, ctx->hcand.p, fp4 ? ctx->himg.p : nullptr);
    gdca_launch_weights(s, (const int32_t *)ctx->hcnt.p, M, gdca_fix_shift(M), (int32_t *)ctx->nk.p,
                        (double *)ctx->W.p, (unsigned long long *)ctx->Wfix.p);
    // (Meff: an exact integer sum by one workgroup, microseconds.  Rounds 1-4 summed left to right in f64 -- 0.3 ms of dependent adds at
    // M = 50 000 -- on a side stream of the context, joined before the first consumer.)
    gdca_launch_meff(s, (const double *)ctx->W.p, M, sc);
    return check_launch(ctx, "weights");
}

// stage 3: Pi, pair tallies.  mode 0 -> Pij_true (ld) ; mode 1 -> covariance (ld)
// want_norm1 (mode 1): the fused path's first build -- sc->pi_max, and sc->mat_norm1 = ||C||_1 where the screen needs it (k_cov_norm1)
static gdca_status tally_stage(gdca_ctx *ctx, const int8_t *Zd, int N, int M, int q, const double *Meff_dev, double pc,
                               int mode, double *Pi_true_out, double *out, size_t ld, bool want_norm1 = false)
{
    hipStream_t s = ctx->stream;
    const int sdim = q - 1, n = N * sdim;
    const int shift = gdca_fix_shift(M);
    const int TJ = gdca_tally_tj(q, ctx->tune.tally_tj);
    CHK(ensure(ctx, ctx->Zt, (size_t)N * M));
    CHK(ensure(ctx, ctx->Zp, (size_t)round_up(N, 64) * M + 64));
    CHK(ensure(ctx, ctx->Pifix, (size_t)N * 32 * sizeof(unsigned long long)));
    CHK(ensure(ctx, ctx->Pipc, (size_t)n * sizeof(double)));
    gdca_launch_transpose_i8(s, Zd, (int8_t *)ctx->Zt.p, N, M);
    gdca_launch_colblock(s, Zd, (int8_t *)ctx->Zp.p, N, M, TJ);
    gdca_fill_async(s, ctx->Pifix.p, 0, (size_t)N * 32 * sizeof(unsigned long long));
    gdca_launch_pi_tally(s, Zd, (const unsigned long long *)ctx->Wfix.p, (unsigned long long *)ctx->Pifix.p, N, M, q,
                         (gdca_dev_scalars *)ctx->sc.p);
    gdca_launch_pi_finalize(s, (const unsigned long long *)ctx->Pifix.p, N, q, shift, Meff_dev, pc, Pi_true_out,
                            (double *)ctx->Pipc.p, want_norm1 ? &((gdca_dev_scalars *)ctx->sc.p)->pi_max : nullptr);
    const bool tm = want_norm1 && ctx->timing && ctx->n_ev >= 18;  // (the fused path's first build: its own device time, gdca_stats.ms_pair_tally)
    if (tm) CHK(mark(ctx, 9));
    gdca_launch_pair_tally(s, (const int8_t *)ctx->Zp.p, (const int8_t *)ctx->Zt.p,
                           (const unsigned long long *)ctx->Wfix.p, N, M, q, shift, Meff_dev, pc,
                           (const double *)ctx->Pipc.p, mode, out, ld, TJ);
    if (tm) CHK(mark(ctx, 10));
    // ||C||_1 for the refinement screen, where the bound that costs nothing (2 N pi_max) does not settle it (k_cov_norm1)
    // (not even launched where the answer is known on the host: pi_max <= (1 - pc) + pc / q whatever the alignment)
    const double pi_cap = (1.0 - pc) + pc / (double)q;
    const bool settled = pc > 0.0 && 2.0 * (double)N * pi_cap * (double)q * (double)q / pc <= ctx->tune.refine_cond && ctx->tune.refine != 1;
    if (want_norm1 && !settled)
        gdca_launch_cov_norm1(s, out, ld, N, q, pc, ctx->tune.refine_cond, (gdca_dev_scalars *)ctx->sc.p, ctx->tune.refine == 1);
    return check_launch(ctx, "tally");
}

// the workspace of one inverse: 2 x 4 panels G and H (pivot groups of up to four blocks, double-buffered by group parity), one
// 128 x 128 pivot inverse, four 512 x 512 scratch matrices for a group's diagonal super-block, the sweep's flags and item table
static gdca_status inverse_job(gdca_ctx *ctx, int n, int n_pad, gdca_inverse_job *job)
{
    const size_t pbytes = (size_t)n_pad * GDCA_TILE * sizeof(double);
    const int nblk = n_pad / GDCA_TILE;
    const size_t sg = (size_t)4 * GDCA_TILE * 4 * GDCA_TILE * sizeof(double);
    const size_t fbytes = gdca_inverse_flag_bytes(n_pad);
    CHK(ensure(ctx, ctx->G, (size_t)8 * pbytes));
    CHK(ensure(ctx, ctx->H, (size_t)8 * pbytes));
    CHK(ensure(ctx, ctx->P, (size_t)GDCA_TILE * GDCA_TILE * sizeof(double)));
    CHK(ensure(ctx, ctx->Sg, 4 * sg + fbytes + (size_t)3 * (nblk + 2) * sizeof(int)));
    if (ctx->item0_cap < 3 * (nblk + 2)) {
        if (ctx->item0_host) HIPCHK(hipHostFree(ctx->item0_host));
        ctx->item0_host = nullptr;
        ctx->item0_cap = 0;
        HIPCHK(hipHostMalloc((void **)&ctx->item0_host, (size_t)3 * (nblk + 2) * sizeof(int), hipHostMallocDefault));
        ctx->item0_cap = 3 * (nblk + 2);
    }
    gdca_inverse_ws &ws = job->ws;
    for (int w = 0; w < 8; ++w) {
        ws.G[w] = (double *)((char *)ctx->G.p + (size_t)w * pbytes);
        ws.H[w] = (double *)((char *)ctx->H.p + (size_t)w * pbytes);
    }
    ws.P = (double *)ctx->P.p;
    ws.Sg[0] = (double *)ctx->Sg.p;
    ws.Sg[1] = (double *)((char *)ctx->Sg.p + sg);
    ws.Pg[0] = (double *)((char *)ctx->Sg.p + 2 * sg);
    ws.Pg[1] = (double *)((char *)ctx->Sg.p + 3 * sg);
    ws.flags = (unsigned *)((char *)ctx->Sg.p + 4 * sg);
    ws.flags_bytes = fbytes;
    ws.item0_dev = (int *)((char *)ctx->Sg.p + 4 * sg + fbytes);
    ws.item0_host = ctx->item0_host;
    {
        void *dp = nullptr;
        ws.item0_host_dev = hipHostGetDevicePointer(&dp, ctx->item0_host, 0) == hipSuccess ? (const int *)dp : nullptr;
    }
    ws.update_cus = ctx->ncu;
    job->A = (double *)ctx->A.p;
    job->n_pad = n_pad;
    job->n_real = n;
    job->sc = (gdca_dev_scalars *)ctx->sc.p;
    job->tune = &ctx->tune;
    job->doomed = (ctx->tune.sweep_debug & 32) != 0 && ctx->pend_attempt == 0;
    return GDCA_OK;
}

static gdca_status poison_inverse_ws(gdca_ctx *ctx, hipStream_t s)
{
    HIPCHK(hipMemsetAsync(ctx->G.p, 0xFF, ctx->G.cap, s));
    HIPCHK(hipMemsetAsync(ctx->H.p, 0xFF, ctx->H.cap, s));
    HIPCHK(hipMemsetAsync(ctx->P.p, 0xFF, ctx->P.cap, s));
    HIPCHK(hipMemsetAsync(ctx->Sg.p, 0xFF, (size_t)4 * 4 * GDCA_TILE * 4 * GDCA_TILE * sizeof(double), s));
    return GDCA_OK;
}

static gdca_status inverse_stage(gdca_ctx *ctx, int n, int n_pad, bool timed, int *n_upd, double *upd_flops)
{
    gdca_inverse_job job;
    CHK(inverse_job(ctx, n, n_pad, &job));
    hipEvent_t *uev = nullptr;
    int max_ev = 0;
    if (timed) {
        CHK(need_events(ctx, 18));
        uev = ctx->ev + 16;
        max_ev = 2;
    }
    // tests (option SWEEP_DEBUG): bit 4 = the panel / scratch buffers are poisoned with NaNs first (an item that reads a buffer
    // before its producer wrote it can then not pass on the leftovers of an earlier, identical run); bit 3 = the merged kernel
    // carries this single inverse
    if (ctx->tune.sweep_debug & 16) CHK(poison_inverse_ws(ctx, ctx->stream));
    if (ctx->tune.sweep_debug & 8) {
        double fl = 0.0;
        gdca_launch_spd_inverse_merged(ctx->stream, &job, 1, uev, max_ev, &fl);
        if (n_upd) *n_upd = 1;
        if (upd_flops) *upd_flops = fl;
        return check_launch(ctx, "spd_inverse_merged");
    }
    gdca_launch_spd_inverse(ctx->stream, job, uev, max_ev, n_upd, upd_flops);
    return check_launch(ctx, "spd_inverse");
}

static gdca_status score_stage(gdca_ctx *ctx, int N, int sdim, int n_pad, int score, int apc, double *S_dev, bool time_fn = false)
{
    hipStream_t s = ctx->stream;
    if (score == GDCA_SCORE_DI) {
        CHK(ensure(ctx, ctx->Tws, gdca_di_ws_bytes(N, sdim)));
        gdca_launch_di(s, (const double *)ctx->A.p, (size_t)n_pad, (const double *)ctx->Ld.p, N, sdim, S_dev,
                       (double *)ctx->Tws.p, (gdca_dev_scalars *)ctx->sc.p);
    } else {
        const bool tm = time_fn && ctx->n_ev >= 18;  // (a timed fused run: gdca_stats.ms_fn)
        if (tm) CHK(mark(ctx, 7));
        gdca_launch_fn(s, (const double *)ctx->A.p, (size_t)n_pad, N, sdim, S_dev, ctx->ncu);
        if (tm) CHK(mark(ctx, 8));
        if (time_fn) ctx->pend_fn_timed = tm;
    }
    if (apc) {
        CHK(ensure(ctx, ctx->colsum, (size_t)N * sizeof(double)));
        gdca_launch_apc(s, S_dev, N, (double *)ctx->colsum.p);
    }
    return check_launch(ctx, "score");
}

static gdca_status validate(gdca_ctx *ctx, int N, int M, int q)
{
    if (!ctx) return GDCA_EINVAL;
    if (N < 1 || M < 1) return fail(ctx, GDCA_EINVAL, "invalid alignment size%s%s", "", "");
    if (q < 2 || q > GDCA_MAXQ) return fail(ctx, GDCA_EINVAL, "parameter q is too big (max 31 is allowed)%s%s", "", "");
    if ((long long)N * (q - 1) > GDCA_MAX_N) return fail(ctx, GDCA_EINVAL, "N*(q-1) is larger than GDCA_MAX_N%s%s", "", "");
    return GDCA_OK;
}

static gdca_status fetch_scalars(gdca_ctx *ctx)
{
    HIPCHK(hipMemcpyAsync(ctx->sc_host, ctx->sc.p, sizeof(gdca_dev_scalars), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GDCA_OK;
}

// An enqueued, not yet collected run (gdca_run_dev_async) owns the ctx workspace: scalars, flags, the pinned item table
// and the big buffers.  Every other entry point that touches the workspace refuses to run until it has been collected.
static gdca_status not_pending(gdca_ctx *ctx)
{
    if (ctx->pending)
        return fail(ctx, GDCA_EINVAL, "a run is still enqueued on this context: call gdca_run_collect first%s%s", "", "");
    return GDCA_OK;
}

// ||X||_1 of the inverse the sweep has just left in ctx->A (sc->inv_norm1): what the collect decides a refinement on
static gdca_status inverse_norm_stage(gdca_ctx *ctx, int n, int n_pad)
{
    if (ctx->tune.refine == 0) return GDCA_OK;
    CHK(ensure(ctx, ctx->normws, (size_t)n_pad * sizeof(double)));
    gdca_launch_inverse_norm1(ctx->stream, (const double *)ctx->A.p, n_pad, n, (double *)ctx->normws.p,
                              &((gdca_dev_scalars *)ctx->sc.p)->inv_norm1);
    return check_launch(ctx, "inverse_norm1");
}

// The screen of the fused path costs ordinary families nothing: the covariance with pseudocount pc is that of a MIXTURE -- weight 1 - pc
// on the reweighted alignment, weight pc on independent uniform columns -- so  C >= pc Cov_uniform  in the positive-definite order, and
// Cov_uniform = blockdiag(I / q - 1 1^T / q^2) (q - 1 states a column) has smallest eigenvalue 1 / q^2:
//       lambda_min(C) >= pc / q^2,     cond_2(C) <= ||C||_1 q^2 / pc
// (attained on every alignment tried: a column without gaps makes the bound sharp).  For ||C||_1 two figures: 2 N pi_max, which the
// single-site frequencies give away (k_pi_finalize), and the measured norm (64 .. 94 on the reference's `large` data) where the first
// is not enough -- k_cov_norm1 decides that on the device with this very formula, before the sweep overwrites C.  At the
// pseudocounts gDCA is used with the bound is 1e4 .. 1e5 and the run pays nothing; beyond the threshold the collect measures ||X||_1
// (one pass over the lower triangle) and decides on kappa_1 = ||C||_1 ||X||_1 like the operator-level entry.  +inf where there is no
// bound (pc = 0).
static double cond_bound(const gdca_dev_scalars &h, double pc, int q, int N)
{
    const double c1 = h.mat_norm1 > 0.0 ? h.mat_norm1 : 2.0 * (double)N * h.pi_max;
    if (!(c1 > 0.0) || !(pc > 0.0)) return HUGE_VAL;
    return c1 * (double)q * (double)q / pc;
}

static bool wants_refinement(const gdca_ctx *ctx, const gdca_dev_scalars &h)
{
    if (h.info != 0 || ctx->tune.refine == 0) return false;
    if (ctx->tune.refine == 1) return true;
    if (!(h.inv_norm1 > 0.0)) return false;  // not measured: the a-priori bound on cond(C) was below the threshold
    // kappa_1 = ||C||_1 ||X||_1 (||C||_1: the covariance build's, or the caller's matrix at the operator-level entry)
    return h.inv_norm1 * (h.mat_norm1 > 0.0 ? h.mat_norm1 : 1.0) > ctx->tune.refine_cond;
}

// The reference's own factorisation as the last resort (option CHOLESKY: 0 never, 1 where the sweep gave up, 2 always [tests]):
// the sweep reported a non-positive pivot -- beyond cond ~1e10 that can be its own rounding, LAPACK's dpotrf still factors such
// matrices -- or its Newton-Schulz step could not converge.  `h`: the scalars as fetched after the sweep / the refinement.
static bool wants_cholesky(const gdca_ctx *ctx, const gdca_dev_scalars &h, int refined)
{
    if (ctx->tune.cholesky == 0 || h.bad_symbol || h.info == INT32_MIN) return false;
    return ctx->tune.cholesky == 2 || h.info > 0 || refined < 0;
}

// ctx->C2 holds the matrix (n_pad x n_pad, identity padding, at least its lower block triangle): dpotrf + dpotri by blocks, -inverse
// into ctx->A where the sweep would have left it; sc->info becomes dpotrf's
static gdca_status cholesky_stage(gdca_ctx *ctx, int n, int n_pad)
{
    gdca_dev_scalars *sc = (gdca_dev_scalars *)ctx->sc.p;
    const size_t mat = (size_t)n_pad * n_pad * sizeof(double);
    CHK(ensure(ctx, ctx->B0, mat));
    CHK(ensure(ctx, ctx->Rt, mat));
    CHK(ensure(ctx, ctx->Wd, (size_t)(n_pad / GDCA_TILE) * GDCA_TILE * GDCA_TILE * sizeof(double)));
    HIPCHK(hipMemsetAsync(&sc->info, 0, sizeof(int), ctx->stream));
    gdca_launch_cholesky_inverse(ctx->stream, (double *)ctx->C2.p, (double *)ctx->B0.p, (double *)ctx->Rt.p, (double *)ctx->Wd.p, (double *)ctx->A.p,
                                 n_pad, n, sc);
    return check_launch(ctx, "cholesky_inverse");
}

static gdca_status begin(gdca_ctx *ctx)
{
    CHK(not_pending(ctx));
    HIPCHK(hipSetDevice(ctx->device));
    CHK(ensure(ctx, ctx->sc, sizeof(gdca_dev_scalars)));
    gdca_fill_async(ctx->stream, ctx->sc.p, 0, sizeof(gdca_dev_scalars));
    ctx->pend_attempt = 0;
    ctx->pend_rescored = false;
    return GDCA_OK;
}

static gdca_status run_inverse(gdca_ctx *ctx);

extern "C" {

gdca_status gdca_run_collect(gdca_ctx *ctx, gdca_stats *st)
{
    if (!ctx) return GDCA_EINVAL;
    if (!ctx->pending) return fail(ctx, GDCA_EINVAL, "no enqueued run to collect%s%s", "", "");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->pending = false;
    ctx->rank_pending = false;  // (collected through this entry, a ranked run's ranking is given up: its arrays are scratch of the next run)
    if (ctx->sc_published) {
        // the run's last kernel wrote the scalars to sc_host: no copy of the runtime's -- a kernel -- behind another context's sweep
        ctx->sc_published = false;
        HIPCHK(hipStreamSynchronize(ctx->stream));
    } else {
        CHK(fetch_scalars(ctx));
    }
    while (ctx->sc_host->info == INT32_MIN && !ctx->sc_host->bad_symbol && ctx->pend_attempt < ctx->tune.sweep_retries) {
        // The sweep's watchdog ended the launch (k_inverse.hip, spin_until: in practice a launch that did not get all its workgroups
        // back after the driver had taken the device's queues off the hardware).  Nothing is wrong with the matrix: the covariance
        // once more from the tallies -- the sweep works in place --, the inverse again as a launch of its own behind the pipeline's
        // gate, the scores again.  A launch that fails option SWEEP_RETRIES times in a row (default 2) is reported as GDCA_EHIP.
        gdca_dev_scalars *sc = (gdca_dev_scalars *)ctx->sc.p;
        const int N = ctx->pend_N, M = ctx->pend_M, q = ctx->pend_q, n_pad = ctx->pend_npad, n = ctx->pend_n;
        ++ctx->pend_attempt;
        CHK(tally_stage(ctx, ctx->pend_Z, N, M, q, &sc->Meff, ctx->pend_p.pseudocount, 1, nullptr, (double *)ctx->A.p, (size_t)n_pad));
        gdca_launch_pad_identity(ctx->stream, (double *)ctx->A.p, n, n_pad);
        HIPCHK(hipMemsetAsync(&sc->info, 0, sizeof(int), ctx->stream));
        HIPCHK(hipMemsetAsync(&sc->di_noconv, 0, sizeof(int), ctx->stream));
        CHK(run_inverse(ctx));
        CHK(score_stage(ctx, N, q - 1, n_pad, ctx->pend_p.score, ctx->pend_p.apc, ctx->pend_S));
        if (ctx->pend_timed && !ctx->pend_stamped) CHK(mark(ctx, 5));
        CHK(fetch_scalars(ctx));
        ctx->pend_rescored = true;
    }
    if (ctx->tune.refine != 0 && ctx->sc_host->info == 0 && !ctx->sc_host->bad_symbol &&
        (ctx->tune.refine == 1 || cond_bound(*ctx->sc_host, ctx->pend_p.pseudocount, ctx->pend_q, ctx->pend_N) > ctx->tune.refine_cond)) {
        CHK(inverse_norm_stage(ctx, ctx->pend_n, ctx->pend_npad));   // cond(C) may be beyond the threshold: ||X||_1 itself
        CHK(fetch_scalars(ctx));
    }
    if (wants_refinement(ctx, *ctx->sc_host) && !ctx->sc_host->bad_symbol) {
        // The inverse looks ill-conditioned (||X||_1 beyond the threshold): the block sweep's error grows like cond^2, so it gets
        // one Newton-Schulz step against the covariance -- built again from the tallies: the sweep worked in place -- and the
        // scores are computed again from the refined inverse.  Synchronous and several times the cost of the run itself: a path
        // for rare inputs (pseudocounts far below the 0.2 .. 0.8 gDCA is used with), taken here so that the enqueue side stays lean.
        gdca_dev_scalars *sc = (gdca_dev_scalars *)ctx->sc.p;
        const int N = ctx->pend_N, M = ctx->pend_M, q = ctx->pend_q, n_pad = ctx->pend_npad, n = ctx->pend_n;
        const size_t mat = (size_t)n_pad * n_pad * sizeof(double);
        CHK(ensure(ctx, ctx->C2, mat));
        CHK(ensure(ctx, ctx->B0, mat));
        CHK(ensure(ctx, ctx->Rt, mat));
        CHK(tally_stage(ctx, ctx->pend_Z, N, M, q, &sc->Meff, ctx->pend_p.pseudocount, 1, nullptr, (double *)ctx->C2.p, (size_t)n_pad));
        gdca_launch_pad_identity(ctx->stream, (double *)ctx->C2.p, n, n_pad);
        gdca_launch_newton_schulz(ctx->stream, (double *)ctx->A.p, (const double *)ctx->C2.p, (double *)ctx->B0.p, (double *)ctx->Rt.p, n_pad,
                                  &sc->ns_resid);
        CHK(check_launch(ctx, "newton_schulz"));
        HIPCHK(hipMemsetAsync(&sc->di_noconv, 0, sizeof(int), ctx->stream));
        CHK(score_stage(ctx, N, q - 1, n_pad, ctx->pend_p.score, ctx->pend_p.apc, ctx->pend_S));
        CHK(fetch_scalars(ctx));
        // the step squares I - X0 C: with that residual at one or beyond (cond(C) past ~1e10: the sweep's own error is of order one
        // there) it cannot have converged, and the caller is told so instead of being handed the result as if it were refined
        ctx->pend_refined = ctx->sc_host->ns_resid < 1.0 ? 1 : -1;
    }
    if (wants_cholesky(ctx, *ctx->sc_host, ctx->pend_refined)) {
        // the sweep gave up on this covariance (a non-positive pivot, or a refinement that cannot converge): once more the
        // reference's way -- the covariance from the tallies again, blocked dpotrf + dpotri, the scores from that inverse.  Its
        // verdict on positive definiteness is LAPACK's.
        gdca_dev_scalars *sc = (gdca_dev_scalars *)ctx->sc.p;
        const int N = ctx->pend_N, M = ctx->pend_M, q = ctx->pend_q, n_pad = ctx->pend_npad, n = ctx->pend_n;
        CHK(ensure(ctx, ctx->C2, (size_t)n_pad * n_pad * sizeof(double)));
        CHK(tally_stage(ctx, ctx->pend_Z, N, M, q, &sc->Meff, ctx->pend_p.pseudocount, 1, nullptr, (double *)ctx->C2.p, (size_t)n_pad));
        gdca_launch_pad_identity(ctx->stream, (double *)ctx->C2.p, n, n_pad);
        CHK(cholesky_stage(ctx, n, n_pad));
        CHK(inverse_norm_stage(ctx, n, n_pad));
        HIPCHK(hipMemsetAsync(&sc->di_noconv, 0, sizeof(int), ctx->stream));
        CHK(score_stage(ctx, N, q - 1, n_pad, ctx->pend_p.score, ctx->pend_p.apc, ctx->pend_S));
        CHK(fetch_scalars(ctx));
        ctx->pend_refined = 2;
    }
    const gdca_dev_scalars &h = *ctx->sc_host;
    hipEvent_t *ev = ctx->ev;
    if (st) {
        memset(st, 0, sizeof(*st));
        st->theta = h.theta;
        st->Meff = h.Meff;
        st->pair_identity_sum = h.pair_sum;
        st->thresh = h.thresh;
        st->info = h.info;
        st->N = ctx->pend_N;
        st->M = ctx->pend_M;
        st->q = ctx->pend_q;
        st->n = ctx->pend_n;
        st->n_pad = ctx->pend_npad;
        st->update_launches = ctx->pend_nupd;
        st->inverse_batch = ctx->pend_batch;
        st->refined = ctx->pend_refined;
        st->sweep_retries = ctx->pend_attempt;
        st->inverse_norm1 = h.inv_norm1;
        st->matrix_norm1 = h.mat_norm1;
        st->cond_bound = ctx->tune.refine != 0 ? cond_bound(h, ctx->pend_p.pseudocount, ctx->pend_q, ctx->pend_N) : 0.0;
        st->inverse_flops = inverse_flops_model((double)ctx->pend_n);
        st->update_flops = ctx->pend_upd_flops;
        st->sweep_ghz = h.sweep_ticks ? (double)h.sweep_cycles / (double)h.sweep_ticks * 0.1 : 0.0;
        if (ctx->pend_timed) {
            // device time between two stage boundaries: HIP events of this context, or -- a run whose kernels went out as batched grids --
            // the 100 MHz time stamps kernels of the batch wrote into its scalars (slot = the event's index)
            bool tfail = false;
            auto between = [&](int a, int b) -> double {
                if (ctx->pend_stamped) return (double)(long long)(h.stamp[b] - h.stamp[a]) * 1e-5;
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev[a], ev[b]) != hipSuccess) tfail = true;
                return ms;
            };
            // (stages issued as batched grids carry the whole batch: a member reports its share, as for a merged inverse)
            const double fshare = 1.0 / (double)(ctx->pend_front_batch > 0 ? ctx->pend_front_batch : 1);
            st->ms_total = between(0, 5);
            st->ms_theta = between(0, 1) * fshare;
            st->ms_weights = between(1, 2) * fshare;
            st->ms_covariance = between(2, 3) * fshare;
            // (a family whose inverse shared a merged launch with others reports its share of that launch: the launch's time
            // divided by the families it carried -- the sum over the members is the launch)
            const double share = 1.0 / (double)(ctx->pend_batch > 0 ? ctx->pend_batch : 1);
            st->ms_score = between(11, 5) / (double)(ctx->pend_score_batch > 0 ? ctx->pend_score_batch : 1);
            if (ctx->pend_inv_stamped) {
                st->ms_inverse = (double)(long long)(h.stamp[4] - h.stamp[6]) * 1e-5 * share;
                st->ms_inverse_update = (double)(long long)(h.stamp[17] - h.stamp[16]) * 1e-5 * share;
            } else {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, ev[6], ev[4]));  // from the start of its turn (after any pipeline gate)
                st->ms_inverse = ms * share;
                HIPCHK(hipEventElapsedTime(&ms, ctx->pend_upd_ev[0], ctx->pend_upd_ev[1]));
                st->ms_inverse_update = ms * share;
            }
            if (ctx->pend_fn_timed && ctx->pend_refined == 0) st->ms_fn = between(7, 8) / (double)(ctx->pend_score_batch > 0 ? ctx->pend_score_batch : 1);
            if (ctx->pend_tally_timed && ctx->pend_refined == 0) st->ms_pair_tally = between(9, 10) * fshare;
            if (tfail) return fail(ctx, GDCA_EHIP, "hipEventElapsedTime%s%s", "", "");
        }
    }
    if (h.bad_symbol) return fail(ctx, GDCA_EINVAL, "alignment holds a symbol outside 1..q%s%s", "", "");
    if (h.info == INT32_MIN)  // the sweep kernel's watchdog (k_inverse.hip, spin_until): a dependency wait ran out of time
        return fail(ctx, GDCA_EHIP, "SPD inverse aborted: a dependency wait inside the sweep kernel timed out%s%s", "", "");
    if (h.info != 0) return fail(ctx, GDCA_ENOTPD, "covariance matrix is not positive definite%s%s", "", "");
    if (h.di_noconv != 0) {
        if (st) st->info = -h.di_noconv;
        return fail(ctx, GDCA_ENOCONV, "eigenvalue iteration of a DI block did not converge%s%s", "", "");
    }
    return GDCA_OK;
}

}  // extern "C"

// The three phases of one run, enqueued on `s` (the ctx's own stream, or the leader's when several families are batched by
// phase): front end (theta, reweighting, tallies, covariance), SPD inverse, scores.
static gdca_status run_check_args(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, int32_t q, const gdca_params *p,
                                  double *S_dev)
{
    CHK(validate(ctx, N, M, q));
    if (!Z_dev || !S_dev || !p) return fail(ctx, GDCA_EINVAL, "null pointer%s%s", "", "");
    if (!(p->pseudocount >= 0.0 && p->pseudocount <= 1.0))
        return fail(ctx, GDCA_EINVAL, "invalid pseudocount value (must be between 0 and 1)%s%s", "", "");
    if (!(p->theta <= 1.0)) return fail(ctx, GDCA_EINVAL, "invalid theta value%s%s", "", "");
    if (p->score != GDCA_SCORE_FROB && p->score != GDCA_SCORE_DI)
        return fail(ctx, GDCA_EINVAL, "invalid score value%s%s", "", "");
    return GDCA_OK;
}

static gdca_status run_front(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, int32_t q, const gdca_params *p)
{
    CHK(begin(ctx));  // (one run may be outstanding per ctx: its scalars and events would be overwritten -> GDCA_EINVAL)
    hipStream_t s = ctx->stream;
    const int sdim = q - 1, n = N * sdim, n_pad = round_up(n, GDCA_TILE);
    const bool timed = ctx->timing;
    if (timed) CHK(need_events(ctx, 18));
    gdca_dev_scalars *sc = (gdca_dev_scalars *)ctx->sc.p;

    if (timed) CHK(mark(ctx, 0));
    CHK(weights_stage(ctx, Z_dev, N, M, q, p->theta, -1, false, timed ? 1 : -1));
    if (timed) CHK(mark(ctx, 2));

    CHK(ensure(ctx, ctx->A, (size_t)n_pad * n_pad * sizeof(double)));
    CHK(tally_stage(ctx, Z_dev, N, M, q, &sc->Meff, p->pseudocount, 1, nullptr, (double *)ctx->A.p, (size_t)n_pad, ctx->tune.refine != 0));
    gdca_launch_pad_identity(s, (double *)ctx->A.p, n, n_pad);
    if (p->score == GDCA_SCORE_DI) {
        CHK(ensure(ctx, ctx->Dblk, (size_t)N * sdim * sdim * sizeof(double)));
        CHK(ensure(ctx, ctx->Ld, (size_t)N * sdim * sdim * sizeof(double)));
        gdca_launch_save_diag_blocks(s, (const double *)ctx->A.p, (size_t)n_pad, N, sdim, (double *)ctx->Dblk.p);
        gdca_launch_diag_chol(s, (const double *)ctx->Dblk.p, N, sdim, (double *)ctx->Ld.p);
    }
    CHK(check_launch(ctx, "covariance"));
    if (timed) CHK(mark(ctx, 3));
    ctx->pend_timed = timed;
    ctx->pend_stamped = ctx->stamped;
    ctx->pend_front_batch = 1;
    ctx->pend_score_batch = 1;
    ctx->pend_tally_timed = timed && ctx->tune.refine != 0;
    ctx->pend_fn_timed = false;
    ctx->pend_Z = Z_dev;
    ctx->pend_p = *p;
    ctx->pend_refined = 0;
    ctx->pend_N = N;
    ctx->pend_M = M;
    ctx->pend_q = q;
    ctx->pend_n = n;
    ctx->pend_npad = n_pad;
    return GDCA_OK;
}

static gdca_status run_inverse(gdca_ctx *ctx)
{
    hipStream_t s = ctx->stream;
    const bool timed = ctx->pend_timed;
    hipEvent_t *ev = ctx->ev;
    int n_upd = 0;
    double upd_flops = 0.0;
    if (ctx->gate && ctx->gate->armed) HIPCHK(hipStreamWaitEvent(s, ctx->gate->ev[ctx->gate->last], 0));
    if (timed) HIPCHK(hipEventRecord(ev[6], s));  // start of this family's turn on the MFMA pipe
    CHK(inverse_stage(ctx, ctx->pend_n, ctx->pend_npad, timed, &n_upd, &upd_flops));
    if (timed) HIPCHK(hipEventRecord(ev[4], s));
    if (ctx->gate) {
        gdca_gate *g = ctx->gate;
        HIPCHK(hipEventRecord(g->ev[g->next], s));
        g->last = g->next;
        g->next = (g->next + 1) & 3;
        g->armed = 1;
    }
    ctx->pend_nupd = n_upd;
    ctx->pend_upd_flops = upd_flops;
    ctx->pend_batch = 1;
    ctx->pend_inv_stamped = false;
    if (timed) {
        ctx->pend_upd_ev[0] = ev[16];
        ctx->pend_upd_ev[1] = ev[17];
    }
    return GDCA_OK;
}

// one launch that stamps slot a (and b, if >= 0) of every given context's scalars (the batched k_stamp)
static void stamp_members(hipStream_t s, gdca_ctx *const *mem, int K, int a, int b)
{
    gdca_recorder *&act = gdca_recorder::active();
    gdca_recorder *outer = act;
    if (outer) (void)outer->flush();
    gdca_recorder rec;
    rec.begin(s, K);
    for (int k = 0; k < K; ++k) {
        rec.member(k);
        gdca_launch_stamp(s, (gdca_dev_scalars *)mem[k]->sc.p, a, b);
    }
    (void)rec.end();
    act = outer;
}

// The inverses of K members of a phase batch as ONE merged launch on the batch's stream (k_sweep_merged: small matrices, which
// leave most of the chip idle when they run alone).  Every member keeps its own workspace, flags, scalars and events.
static gdca_status run_inverse_merged(gdca_ctx *lead, gdca_ctx *const *mem, int K)
{
    gdca_ctx *ctx = mem[0];
    hipStream_t s = ctx->stream;
    gdca_inverse_job jobs[8];
    gdca_tuning tun[8];  // a member's schedule switches are its own context's, the switches of the merged launch the leader's
    double flops[8];
    if (K > gdca_inverse_max_merge() || K > 8) return fail(ctx, GDCA_EINVAL, "too many members in a merged inverse%s%s", "", "");
    bool timed = true, stamped = true;
    // (the members of a batch are normally peers of ONE pipeline: one wait and one record per distinct gate, not per member)
    gdca_gate *gates[8];
    int n_gates = 0;
    for (int k = 0; k < K; ++k) {
        timed = timed && mem[k]->pend_timed;
        stamped = stamped && mem[k]->pend_stamped;
        gdca_gate *g = mem[k]->gate;
        bool seen = !g;
        for (int j = 0; j < n_gates && !seen; ++j) seen = gates[j] == g;
        if (!seen) {
            gates[n_gates++] = g;
            if (g->armed) HIPCHK(hipStreamWaitEvent(s, g->ev[g->last], 0));
        }
    }
    stamped = stamped && timed;
    for (int k = 0; k < K; ++k) {
        if (mem[k]->pend_timed && !stamped) HIPCHK(hipEventRecord(mem[k]->ev[6], s));
        CHK(inverse_job(mem[k], mem[k]->pend_n, mem[k]->pend_npad, &jobs[k]));
        tun[k] = mem[k]->tune;
        tun[k].merge_group = lead->tune.merge_group;
        tun[k].merge_mcus = lead->tune.merge_mcus;
        jobs[k].tune = &tun[k];
        if (lead->tune.sweep_debug & 16) CHK(poison_inverse_ws(mem[k], s));
    }
    // every timed member brackets the launch with events of ITS OWN: members may be collected, enqueued again or destroyed in any
    // order, and a collect must never read another context's events -- or, members of a batch issued as batched grids,
    // with two launches that stamp every member's scalars (2 launches instead of 4 K event records)
    if (stamped)
        stamp_members(s, mem, K, 6, 16);
    else if (timed)
        for (int k = 0; k < K; ++k) {
            CHK(need_events(mem[k], 18));
            HIPCHK(hipEventRecord(mem[k]->ev[16], s));
        }
    gdca_launch_spd_inverse_merged(s, jobs, K, nullptr, 0, flops);
    CHK(check_launch(ctx, "spd_inverse_merged"));
    if (stamped)
        stamp_members(s, mem, K, 17, 4);
    else if (timed)
        for (int k = 0; k < K; ++k) HIPCHK(hipEventRecord(mem[k]->ev[17], s));
    for (int j = 0; j < n_gates; ++j) {
        gdca_gate *g = gates[j];
        HIPCHK(hipEventRecord(g->ev[g->next], s));
        g->last = g->next;
        g->next = (g->next + 1) & 3;
        g->armed = 1;
    }
    for (int k = 0; k < K; ++k) {
        gdca_ctx *m = mem[k];
        if (m->pend_timed && !stamped) HIPCHK(hipEventRecord(m->ev[4], s));
        m->pend_nupd = k == 0 ? 1 : 0;
        m->pend_upd_flops = flops[k];
        m->pend_batch = K;
        // (a member without timing of its own still gets valid events to read: the launch's, or none -- collect reads them only if timed)
        m->pend_timed = m->pend_timed && timed;
        m->pend_inv_stamped = stamped;
        if (timed && !stamped) {
            m->pend_upd_ev[0] = m->ev[16];
            m->pend_upd_ev[1] = m->ev[17];
        }
    }
    return GDCA_OK;
}

static gdca_status run_score(gdca_ctx *ctx, const gdca_params *p, double *S_dev)
{
    ctx->pend_S = S_dev;
    // (the start of THIS run's score stage: in a phase batch the members' stages follow one another behind the shared inverse)
    if (ctx->pend_timed) CHK(mark(ctx, 11));
    CHK(score_stage(ctx, ctx->pend_N, ctx->pend_q - 1, ctx->pend_npad, p->score, p->apc, S_dev, ctx->pend_timed));
    if (ctx->pend_timed) CHK(mark(ctx, 5));
    ctx->sc_published = false;
    if (ctx->sc_host_dev) {
        gdca_launch_publish_scalars(ctx->stream, (const gdca_dev_scalars *)ctx->sc.p, ctx->sc_host_dev);
        CHK(check_launch(ctx, "publish_scalars"));
        ctx->sc_published = true;
    }
    ctx->pending = true;
    return GDCA_OK;
}

// The inverses of the members of a batch, all enqueued on the batch's stream (every member's `stream` points there, pend_n /
// pend_npad are set): those of the small members (up to `merge_blocks` 128-blocks: chain-bound single-block schedules that leave
// most of the chip idle) are carried `merge` at a time by ONE launch (options GDCA_MERGE, GDCA_MERGE_BLOCKS of the leader; bit for
// bit the results of launches of their own), the others run back to back
static gdca_status run_inverses(gdca_ctx *lead, gdca_ctx *const *ctxs, int K)
{
    gdca_ctx *small[64];
    int n_small = 0;
    gdca_status st = GDCA_OK;
    const int merge = std::min(lead->tune.merge, gdca_inverse_max_merge());
    for (int k = 0; k < K && st == GDCA_OK; ++k) {
        if (merge > 1 && ctxs[k]->pend_npad / GDCA_TILE <= lead->tune.merge_blocks)
            small[n_small++] = ctxs[k];
        else
            st = run_inverse(ctxs[k]);
    }
    // A launch is closed when it holds `merge` members or its members together offer MERGE_TILES tile items per update step
    // (sum of nblk^2 / 2: enough work beside a member's chain -- ~250 us per group of four blocks against ~0.2 us per tile item
    // on the whole chip; 20-block matrices go eight to a launch, 47-block ones two).  A lone last member joins the launch
    // before it.
    int start[65], n_grp = 0;
    {
        long long tiles = 0;
        int cnt = 0;
        for (int k = 0; k < n_small; ++k) {
            if (cnt == 0) start[n_grp++] = k;
            const long long nb = small[k]->pend_npad / GDCA_TILE;
            tiles += nb * nb / 2;
            ++cnt;
            if (cnt == merge || tiles >= lead->tune.merge_tiles) {
                cnt = 0;
                tiles = 0;
            }
        }
        start[n_grp] = n_small;
        if (n_grp >= 2 && start[n_grp] - start[n_grp - 1] == 1 && start[n_grp - 1] - start[n_grp - 2] < merge) start[--n_grp] = n_small;
    }
    for (int gi = 0; gi < n_grp && st == GDCA_OK; ++gi) {
        const int at = start[gi], cnt = start[gi + 1] - at;
        // (option SWEEP_DEBUG bit 3, tests: the merged kernel for a single member too)
        st = (cnt == 1 && !(lead->tune.sweep_debug & 8)) ? run_inverse(small[at]) : run_inverse_merged(lead, small + at, cnt);
    }
    return st;
}

extern "C" {

gdca_status gdca_run_dev_async(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, int32_t q,
                               const gdca_params *p, double *S_dev)
{
    CHK(run_check_args(ctx, Z_dev, N, M, q, p, S_dev));
    CHK(run_front(ctx, Z_dev, N, M, q, p));
    CHK(run_inverse(ctx));
    return run_score(ctx, p, S_dev);
}

static gdca_status ranking_stage(gdca_ctx *ctx, const double *S_dev, int N, int sep, long long len, int32_t **ii, int32_t **jj, double **sc);

static gdca_status run_phased(gdca_ctx *const *ctxs, int32_t K, const int8_t *const *Z_dev, const int32_t *N, const int32_t *M, const int32_t *q,
                              const gdca_params *p, double *const *S_dev, int rank_sep, bool *ranked)
{
    if (ranked) *ranked = false;
    if (!ctxs || K < 1 || !Z_dev || !N || !M || !q || !p || !S_dev) return GDCA_EINVAL;
    gdca_ctx *lead = ctxs[0];
    if (!lead) return GDCA_EINVAL;
    for (int k = 0; k < K; ++k) {
        if (!ctxs[k]) return fail(lead, GDCA_EINVAL, "null context in the batch%s%s", "", "");
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return fail(lead, GDCA_EINVAL, "the same context twice in one batch%s%s", "", "");
        if (ctxs[k]->device != lead->device) return fail(lead, GDCA_EINVAL, "contexts of one batch must share a device%s%s", "", "");
        gdca_status vs = not_pending(ctxs[k]);
        if (vs == GDCA_OK) vs = run_check_args(ctxs[k], Z_dev[k], N[k], M[k], q[k], p, S_dev[k]);
        if (vs != GDCA_OK) {
            if (k > 0) {  // the caller reads the leader's last_error
                char msg[sizeof(lead->err)];
                snprintf(msg, sizeof(msg), "member %d: %.400s", k, ctxs[k]->err);
                memcpy(lead->err, msg, sizeof(msg));
            }
            return vs;
        }
    }
    // Phase by phase: K front ends, K inverses back to back, K score stages.  The inverses and the score stages go to the leader's
    // stream.  The front ends (reweighting, tallies, covariance: kernels of a few dozen to a few hundred workgroups each for a small
    // family -- the pair tally of an N = 128 family launches 36 on 256 compute units) run SIDE BY SIDE, each on its member's own
    // stream, and the leader's stream waits for all of them before the first inverse (option PHASED_FRONTS=0: one after the other
    // on the leader's stream, as in round 4: 8 x 0.55 ms in front of 8 x 0.43 ms of merged inverses at config B).  The members keep
    // their own workspaces, scalars and timing events; their streams are restored before returning.
    hipStream_t own[64], use[64];
    if (K > 64) return fail(lead, GDCA_EINVAL, "at most 64 families per batch%s%s", "", "");
    // Round 6, the default (option PHASED_GRIDS != 0): the members' kernels of a KIND as ONE grid (gdca_launch.h).  The front ends are
    // recorded member by member, then issued in lockstep on the leader's stream -- ~25 launches per batch instead of ~25 per member,
    // every one with all members' workgroups on the chip at once --, then the inverses, then the score stages (and, for the ranked
    // entry, the rankings) the same way.  Stage boundaries are time stamps written by kernels of the batch, not events.  Bit for bit
    // the results of the other schedules: a kernel body cannot tell which launch form runs it.
    if (lead->tune.phased_grids != 0 && K > 1) {  // (-1: the rule below; 1 .. 8: that many groups)
        for (int k = 0; k < K; ++k) {
            own[k] = ctxs[k]->stream;
            if (k > 0) (void)hipStreamSynchronize(own[k]);  // nothing of an earlier use is still in flight on the member's own stream
        }
        gdca_status st = GDCA_OK;
        int done_front = 0, failed = -1;
        gdca_recorder rec;
        // the batch's stream: the pipeline's (gdca_gate) where the leader belongs to one, behind whatever its own stream holds
        hipStream_t bs = lead->gate && lead->gate->batch_stream ? lead->gate->batch_stream : lead->stream;
        if (bs != lead->stream && (hipEventRecord(lead->ev_upload, lead->stream) != hipSuccess || hipStreamWaitEvent(bs, lead->ev_upload, 0) != hipSuccess))
            return fail(lead, GDCA_EHIP, "event chain of the batch%s%s", "", "");
        // G groups of members (member k in group k mod G), each group's front ends as batched grids on a stream of its own -- the
        // leader's and the next members' --, the groups side by side: a VALU-bound reweighting of one group runs beside the
        // LDS-atomic-bound pair tallies of another.  One group (everything in lockstep) where the members are small: their kernels
        // are launch-bound and two groups would be twice the launches.
        int G = lead->tune.phased_grids;
        if (G < 0) {
            double work = 0.0;  // symbol compares + tallies of the batch's front ends
            for (int k = 0; k < K; ++k) work += 0.5 * (double)M[k] * (double)M[k] * (double)N[k] + 0.5 * (double)N[k] * (double)N[k] * (double)M[k];
            G = work / K > 2e10 ? 4 : 1;
        }
        G = std::max(1, std::min(std::min(G, (int)K), 8));
        int in_group[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int gi = 0; gi < G && st == GDCA_OK; ++gi) {
            hipStream_t sg = gi == 0 ? bs : own[gi];
            rec.begin(sg, K);
            for (int k = gi; k < K && st == GDCA_OK; k += G) {
                ctxs[k]->stream = sg;
                ctxs[k]->stamped = true;
                rec.member(k);
                st = run_front(ctxs[k], Z_dev[k], N[k], M[k], q[k], p);
                if (st == GDCA_OK) {
                    ++done_front;
                    ++in_group[gi];
                } else {
                    failed = k;
                }
            }
            rec.member(-1);
            if (rec.flush() != hipSuccess && st == GDCA_OK) st = fail(lead, GDCA_EHIP, "a batched front-end launch failed%s%s", "", "");
            // the batch's stream goes on behind this group's front ends
            if (st == GDCA_OK && gi > 0 &&
                (hipEventRecord(ctxs[gi]->ev_batch, sg) != hipSuccess || hipStreamWaitEvent(bs, ctxs[gi]->ev_batch, 0) != hipSuccess))
                st = fail(lead, GDCA_EHIP, "event chain of the batch's front ends%s%s", "", "");
        }
        if (st != GDCA_OK) done_front = 0;  // (nothing of a batch that failed to enqueue is run any further; it is drained below)
        for (int k = 0; k < K; ++k) {
            ctxs[k]->stream = bs;
            if (st == GDCA_OK) ctxs[k]->pend_front_batch = in_group[k % G];
        }
        rec.begin(bs, K);
        if (st == GDCA_OK) st = run_inverses(lead, ctxs, done_front);
        for (int k = 0; k < done_front && st == GDCA_OK; ++k) {
            rec.member(k);
            ctxs[k]->pend_score_batch = done_front;
            st = run_score(ctxs[k], p, S_dev[k]);
            if (st == GDCA_OK && rank_sep > 0) {
                gdca_ctx *m = ctxs[k];
                m->rank_len = gdca_ranking_length(N[k], rank_sep);
                m->rank_sep = rank_sep;
                m->rank_status = m->rank_len > 0 ? ranking_stage(m, S_dev[k], N[k], rank_sep, m->rank_len, &m->rank_i, &m->rank_j, &m->rank_s) : GDCA_OK;
                m->rank_pending = true;
            }
            if (st != GDCA_OK) failed = k;
        }
        rec.member(-1);
        if (rec.end() != hipSuccess && st == GDCA_OK) st = fail(lead, GDCA_EHIP, "a batched score launch failed%s%s", "", "");
        if (ranked && st == GDCA_OK && rank_sep > 0) *ranked = true;
        // the members' collects synchronise THEIR stream: every one of them waits for the batch (one event on the batch's stream)
        bool chained = st == GDCA_OK && hipEventRecord(lead->ev_batch, bs) == hipSuccess;
        for (int k = bs == own[0] ? 1 : 0; k < K && chained; ++k) chained = hipStreamWaitEvent(own[k], lead->ev_batch, 0) == hipSuccess;
        for (int k = 0; k < K; ++k) {
            ctxs[k]->stream = own[k];
            ctxs[k]->stamped = false;
        }
        if (st != GDCA_OK) {
            (void)hipStreamSynchronize(bs);
            for (int k = 0; k < K; ++k) {
                (void)hipStreamSynchronize(own[k]);  // (a group's front ends)
                ctxs[k]->pending = false;
                ctxs[k]->rank_pending = false;
            }
            if (failed > 0 && ctxs[failed]->err[0]) {
                char msg[sizeof(lead->err)];
                snprintf(msg, sizeof(msg), "member %d: %.400s", failed, ctxs[failed]->err);
                memcpy(lead->err, msg, sizeof(msg));
            }
            return st;
        }
        if (!chained) {
            (void)hipStreamSynchronize(bs);
            return fail(lead, GDCA_EHIP, "event chain of the batch%s%s", "", "");
        }
        return GDCA_OK;
    }
    const bool side_by_side = lead->tune.phased_fronts != 0 && K > 1;
    // ... on the streams of the first PHASED_STREAMS members, round robin (default 4, the runtime's hardware queues: with one
    // stream per member the eight front ends of a config-B batch ran two to three wide and started up to 1.4 ms apart,
    // profiles/r05_B_merged8_timeline.log -- streams that share a hardware queue take turns, and the second of a pair started
    // hundreds of microseconds after the first had finished)
    const int W = side_by_side ? std::max(1, std::min((int)K, lead->tune.phased_streams)) : 1;
    for (int k = 0; k < K; ++k) {
        own[k] = ctxs[k]->stream;
        if (k > 0) (void)hipStreamSynchronize(own[k]);  // nothing of an earlier use is still in flight on the member's own stream
    }
    for (int k = 0; k < K; ++k) use[k] = side_by_side ? own[k % W] : lead->stream;
    gdca_status st = GDCA_OK;
    int done_front = 0;
    for (int k = 0; k < K && st == GDCA_OK; ++k) {
        ctxs[k]->stream = use[k];
        st = run_front(ctxs[k], Z_dev[k], N[k], M[k], q[k], p);
        if (st == GDCA_OK && use[k] != lead->stream) {
            // the batch's stream goes on behind this member's front end
            if (hipEventRecord(ctxs[k]->ev_batch, use[k]) != hipSuccess || hipStreamWaitEvent(lead->stream, ctxs[k]->ev_batch, 0) != hipSuccess)
                st = fail(lead, GDCA_EHIP, "event chain of the batch's front ends%s%s", "", "");
        }
        if (st == GDCA_OK) ++done_front;
    }
    for (int k = 0; k < K; ++k) ctxs[k]->stream = lead->stream;
    if (st == GDCA_OK) st = run_inverses(lead, ctxs, done_front);
    if (st == GDCA_OK && side_by_side) {
        // ... and the score stages side by side again, each behind the batch's inverses on the stream its front end ran on
        if (hipEventRecord(lead->ev_upload, lead->stream) != hipSuccess) st = fail(lead, GDCA_EHIP, "event chain of the batch's score stages%s%s", "", "");
        for (int j = 1; j < W && st == GDCA_OK; ++j)
            if (hipStreamWaitEvent(own[j], lead->ev_upload, 0) != hipSuccess) st = fail(lead, GDCA_EHIP, "event chain of the batch's score stages%s%s", "", "");
    }
    bool chained = true;  // every member's own stream waits for what was enqueued for it elsewhere
    for (int k = 0; k < done_front && st == GDCA_OK; ++k) {
        ctxs[k]->stream = use[k];
        st = run_score(ctxs[k], p, S_dev[k]);
        // the members' collects synchronise THEIR stream: make it wait for the stream the member's work went to (one event per member)
        if (st == GDCA_OK && use[k] != own[k] &&
            (hipEventRecord(ctxs[k]->ev_batch, use[k]) != hipSuccess || hipStreamWaitEvent(own[k], ctxs[k]->ev_batch, 0) != hipSuccess))
            chained = false;
    }
    if (st != GDCA_OK) {
        // a member failed to enqueue (allocation, launch): drain what was enqueued -- the batch's stream AND every member's own
        // stream -- and leave nobody half-pending.  The failing member's message goes to the leader, whose last_error the caller reads.
        (void)hipStreamSynchronize(lead->stream);
        for (int k = 0; k < K; ++k) {
            (void)hipStreamSynchronize(own[k]);  // (a front end enqueued side by side)
            ctxs[k]->pending = false;
            if (k > 0 && ctxs[k]->err[0] && k == done_front) {
                char msg[sizeof(lead->err)];
                snprintf(msg, sizeof(msg), "member %d: %.400s", k, ctxs[k]->err);
                memcpy(lead->err, msg, sizeof(msg));
            }
        }
    }
    for (int k = 0; k < K; ++k) ctxs[k]->stream = own[k];
    if (st == GDCA_OK && !chained) {
        // the chain could not be built: the members' collects would not wait for the batch -- wait for it here instead
        (void)hipStreamSynchronize(lead->stream);
        for (int j = 1; j < W; ++j) (void)hipStreamSynchronize(own[j]);
        return fail(lead, GDCA_EHIP, "event chain of the batch%s%s", "", "");
    }
    return st;
}

gdca_status gdca_run_dev_phased(gdca_ctx *const *ctxs, int32_t K, const int8_t *const *Z_dev, const int32_t *N,
                                const int32_t *M, const int32_t *q, const gdca_params *p, double *const *S_dev)
{
    return run_phased(ctxs, K, Z_dev, N, M, q, p, S_dev, 0, nullptr);
}

gdca_status gdca_run_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, int32_t q, const gdca_params *p,
                         double *S_dev, gdca_stats *st)
{
    CHK(gdca_run_dev_async(ctx, Z_dev, N, M, q, p, S_dev));
    return gdca_run_collect(ctx, st);
}

gdca_status gdca_run(gdca_ctx *ctx, const int8_t *Z_host, int32_t N, int32_t M, int32_t q, const gdca_params *p,
                     double *S_host, gdca_stats *st)
{
    CHK(validate(ctx, N, M, q));
    if (!Z_host || !S_host) return fail(ctx, GDCA_EINVAL, "null pointer%s%s", "", "");
    HIPCHK(hipSetDevice(ctx->device));
    CHK(ensure(ctx, ctx->scratch[0], (size_t)N * M));
    CHK(ensure(ctx, ctx->scratch[1], (size_t)N * N * sizeof(double)));
    HIPCHK(hipMemcpyAsync(ctx->scratch[0].p, Z_host, (size_t)N * M, hipMemcpyHostToDevice, ctx->stream));
    gdca_status rs = gdca_run_dev(ctx, (const int8_t *)ctx->scratch[0].p, N, M, q, p, (double *)ctx->scratch[1].p, st);
    if (rs != GDCA_OK) return rs;
    HIPCHK(hipMemcpyAsync(S_host, ctx->scratch[1].p, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GDCA_OK;
}

// ---- compute_ranking on the device (src/GaussDCA.jl:88-99, call :44) ---------------------------------------
static gdca_status ranking_stage(gdca_ctx *ctx, const double *S_dev, int N, int sep, long long len, int32_t **ii, int32_t **jj, double **sc)
{
    CHK(ensure(ctx, ctx->rankws, gdca_ranking_ws_bytes(len)));
    gdca_launch_ranking(ctx->stream, S_dev, N, sep, len, ctx->rankws.p, ii, jj, sc);
    return check_launch(ctx, "ranking");
}

static gdca_status ranking_to_host(gdca_ctx *ctx, long long len, const int32_t *ii, const int32_t *jj, const double *sc, int32_t *i_out, int32_t *j_out,
                                   double *score_out)
{
    HIPCHK(hipMemcpyAsync(i_out, ii, (size_t)len * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(j_out, jj, (size_t)len * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(score_out, sc, (size_t)len * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GDCA_OK;
}

gdca_status gdca_ranking_dev(gdca_ctx *ctx, const double *S_dev, int32_t N, int32_t min_separation, int32_t *i_out, int32_t *j_out, double *score_out)
{
    if (!ctx || !S_dev || N < 1 || N > 65535 || min_separation < 1) return GDCA_EINVAL;
    CHK(not_pending(ctx));
    HIPCHK(hipSetDevice(ctx->device));
    const long long len = gdca_ranking_length(N, min_separation);
    if (len == 0) return GDCA_OK;
    if (!i_out || !j_out || !score_out) return GDCA_EINVAL;
    int32_t *ii, *jj;
    double *sc;
    CHK(ranking_stage(ctx, S_dev, N, min_separation, len, &ii, &jj, &sc));
    return ranking_to_host(ctx, len, ii, jj, sc, i_out, j_out, score_out);
}

gdca_status gdca_run_ranked_async(gdca_ctx *ctx, const int8_t *Z_host, int32_t N, int32_t M, int32_t q, const gdca_params *p, int32_t min_separation)
{
    CHK(validate(ctx, N, M, q));
    CHK(not_pending(ctx));
    if (!Z_host || min_separation < 1) return fail(ctx, GDCA_EINVAL, "null pointer or min_separation < 1%s%s", "", "");
    const long long len = gdca_ranking_length(N, min_separation);
    HIPCHK(hipSetDevice(ctx->device));
    CHK(ensure(ctx, ctx->scratch[0], (size_t)N * M));
    CHK(ensure(ctx, ctx->scratch[1], (size_t)N * N * sizeof(double)));
    // (from pageable memory this copy holds the calling thread until the last byte is staged -- and overlaps whatever another
    // context of the same device is computing meanwhile: the pipelined batch driver's upload of the NEXT family)
    HIPCHK(hipMemcpyAsync(ctx->scratch[0].p, Z_host, (size_t)N * M, hipMemcpyHostToDevice, ctx->stream));
    CHK(upload_done(ctx));  // Z_host is the caller's again when this call returns (gdca.h), pinned or not
    double *S_dev = (double *)ctx->scratch[1].p;
    CHK(gdca_run_dev_async(ctx, (const int8_t *)ctx->scratch[0].p, N, M, q, p, S_dev));
    // the ranking is enqueued behind the scores before anybody waits: one synchronisation for the whole run
    ctx->rank_len = len;
    ctx->rank_sep = min_separation;
    ctx->rank_status = len > 0 ? ranking_stage(ctx, S_dev, N, min_separation, len, &ctx->rank_i, &ctx->rank_j, &ctx->rank_s) : GDCA_OK;
    ctx->rank_pending = true;
    return GDCA_OK;
}

gdca_status gdca_run_ranked_phased_async(gdca_ctx *const *ctxs, int32_t K, const int8_t *const *Z_host, const int32_t *N, const int32_t *M,
                                         const int32_t *q, const gdca_params *p, int32_t min_separation)
{
    if (!ctxs || K < 1 || K > 64 || !Z_host || !N || !M || !q || !p) return GDCA_EINVAL;
    gdca_ctx *lead = ctxs[0];
    if (!lead) return GDCA_EINVAL;
    gdca_ctx *ctx = lead;  // (the error macros' context)
    if (min_separation < 1) return fail(lead, GDCA_EINVAL, "min_separation < 1%s%s", "", "");
    const int8_t *Zd[64];
    double *Sd[64];
    for (int k = 0; k < K; ++k) {
        gdca_ctx *m = ctxs[k];
        if (!m || !Z_host[k]) return fail(lead, GDCA_EINVAL, "null context or alignment in the batch%s%s", "", "");
        gdca_status vs = validate(m, N[k], M[k], q[k]);
        if (vs == GDCA_OK) vs = not_pending(m);
        if (vs != GDCA_OK) {
            if (k > 0) {
                char msg[sizeof(lead->err)];
                snprintf(msg, sizeof(msg), "member %d: %.400s", k, m->err);
                memcpy(lead->err, msg, sizeof(msg));
            }
            return vs;
        }
    }
    HIPCHK(hipSetDevice(lead->device));
    for (int k = 0; k < K; ++k) {
        gdca_ctx *m = ctxs[k];
        gdca_status es = ensure(m, m->scratch[0], (size_t)N[k] * M[k]);
        if (es == GDCA_OK) es = ensure(m, m->scratch[1], (size_t)N[k] * N[k] * sizeof(double));
        if (es != GDCA_OK) return member_error(lead, m, k, es);
        // (each member's upload goes to its own stream; gdca_run_dev_phased waits for the members' streams before it enqueues
        // anything, and the leader's upload sits on the very stream the batch is enqueued on)
        if (hipMemcpyAsync(m->scratch[0].p, Z_host[k], (size_t)N[k] * M[k], hipMemcpyHostToDevice, m->stream) != hipSuccess)
            return fail(lead, GDCA_EHIP, "upload of a member's alignment%s%s", "", "");
        Zd[k] = (const int8_t *)m->scratch[0].p;
        Sd[k] = (double *)m->scratch[1].p;
    }
    // Z_host may be released when this call returns (gdca.h): from pageable memory the copies above are complete by now, from
    // pinned or registered memory they are true DMAs still in flight -- wait for every one of them HERE, whatever path the call
    // takes afterwards (gdca_run_dev_phased synchronises the members' streams too, but not on its early argument checks)
    CHK(upload_done(lead));
    for (int k = 1; k < K; ++k)
        if (hipStreamSynchronize(ctxs[k]->stream) != hipSuccess) return fail(lead, GDCA_EHIP, "upload of a member's alignment%s%s", "", "");
    bool ranked = false;  // (issued as batched grids, the rankings are part of the batch)
    CHK(run_phased(ctxs, K, Zd, N, M, q, p, Sd, min_separation, &ranked));
    // every member's ranking behind its scores, on the member's own stream (which now waits for the batch)
    for (int k = 0; k < K && !ranked; ++k) {
        gdca_ctx *m = ctxs[k];
        m->rank_len = gdca_ranking_length(N[k], min_separation);
        m->rank_sep = min_separation;
        m->rank_status = m->rank_len > 0 ? ranking_stage(m, Sd[k], N[k], min_separation, m->rank_len, &m->rank_i, &m->rank_j, &m->rank_s) : GDCA_OK;
        m->rank_pending = true;
    }
    return GDCA_OK;
}

gdca_status gdca_run_ranked_collect(gdca_ctx *ctx, int32_t *i_out, int32_t *j_out, double *score_out, gdca_stats *st)
{
    if (!ctx) return GDCA_EINVAL;
    if (!ctx->rank_pending) return fail(ctx, GDCA_EINVAL, "no enqueued ranked run to collect%s%s", "", "");
    ctx->rank_pending = false;
    gdca_stats own;
    gdca_status cs = gdca_run_collect(ctx, st ? st : &own);
    if (cs != GDCA_OK) return cs;
    if (ctx->rank_status != GDCA_OK) return ctx->rank_status;
    const long long len = ctx->rank_len;
    if (len == 0) return GDCA_OK;
    if (!i_out || !j_out || !score_out) return fail(ctx, GDCA_EINVAL, "null pointer%s%s", "", "");
    // (a run whose inverse was refined, recomputed or run again at collect time has new scores: rank those)
    if (ctx->pend_refined != 0 || ctx->pend_rescored)
        CHK(ranking_stage(ctx, ctx->pend_S, ctx->pend_N, ctx->rank_sep, len, &ctx->rank_i, &ctx->rank_j, &ctx->rank_s));
    return ranking_to_host(ctx, len, ctx->rank_i, ctx->rank_j, ctx->rank_s, i_out, j_out, score_out);
}

gdca_status gdca_run_ranked(gdca_ctx *ctx, const int8_t *Z_host, int32_t N, int32_t M, int32_t q, const gdca_params *p, int32_t min_separation,
                            int32_t *i_out, int32_t *j_out, double *score_out, gdca_stats *st)
{
    if (ctx && gdca_ranking_length(N, min_separation) > 0 && (!i_out || !j_out || !score_out)) return fail(ctx, GDCA_EINVAL, "null pointer%s%s", "", "");
    CHK(gdca_run_ranked_async(ctx, Z_host, N, M, q, p, min_separation));
    return gdca_run_ranked_collect(ctx, i_out, j_out, score_out, st);
}

// ---- caller-visible device buffers ----------------------------------------------------------------------
// A gdca_dbuf is nothing but an owned HBM allocation: the `_dev` operators below take plain device pointers
// (gdca_dbuf_ptr of such a buffer, or any other device memory of the same GPU, e.g. a torch tensor's data_ptr).

struct gdca_dbuf {
    void *p;
    size_t bytes;
    int device;
};

gdca_status gdca_dbuf_alloc(gdca_ctx *ctx, uint64_t bytes, gdca_dbuf **out)
{
    if (!ctx || !out) return GDCA_EINVAL;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    gdca_dbuf *b = (gdca_dbuf *)calloc(1, sizeof(gdca_dbuf));
    if (!b) return GDCA_ENOMEM;
    const size_t want = bytes ? (size_t)bytes : 16;
    hipError_t e = hipMalloc(&b->p, want);
    if (e != hipSuccess) {
        free(b);
        (void)hipGetLastError();
        return fail(ctx, GDCA_ENOMEM, "hipMalloc failed: %s%s", hipGetErrorString(e), "");
    }
    b->bytes = want;
    b->device = ctx->device;
    *out = b;
    return GDCA_OK;
}

gdca_status gdca_dbuf_free(gdca_dbuf *b)
{
    if (!b) return GDCA_OK;
    (void)hipSetDevice(b->device);
    if (b->p) (void)hipFree(b->p);
    free(b);
    return GDCA_OK;
}

void *gdca_dbuf_ptr(const gdca_dbuf *b)
{
    return b ? b->p : nullptr;
}

uint64_t gdca_dbuf_bytes(const gdca_dbuf *b)
{
    return b ? (uint64_t)b->bytes : 0;
}

gdca_status gdca_dbuf_upload(gdca_ctx *ctx, gdca_dbuf *b, uint64_t offset, const void *host, uint64_t bytes)
{
    if (!ctx || !b || (!host && bytes) || offset + bytes > b->bytes) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    if (bytes) HIPCHK(hipMemcpyAsync((char *)b->p + offset, host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the host buffer may be reused on return
    return GDCA_OK;
}

gdca_status gdca_dbuf_download(gdca_ctx *ctx, const gdca_dbuf *b, uint64_t offset, void *host, uint64_t bytes)
{
    if (!ctx || !b || (!host && bytes) || offset + bytes > b->bytes) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    if (bytes) HIPCHK(hipMemcpyAsync(host, (const char *)b->p + offset, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GDCA_OK;
}

// ---- operator level, device-resident (`_dev`): every matrix argument is a device pointer ----------------------
// Each call enqueues on the ctx stream and returns after the stream has drained only where a host scalar comes
// back (weights: Meff/theta/thresh; spd_inverse: info; di: convergence; frequencies: the symbol-range check); the
// elementwise ones (add_pseudocount, covariance, fn, apc) return right after the launch.

static gdca_status symbols_ok(gdca_ctx *ctx)
{
    if (ctx->sc_host->bad_symbol)
        return fail(ctx, GDCA_EINVAL, "alignment holds a symbol outside 1..q%s%s", "", "");
    return GDCA_OK;
}

gdca_status gdca_pair_identity_sum_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, uint64_t *out)
{
    CHK(validate(ctx, N, M, 2));
    if (!Z_dev || !out) return GDCA_EINVAL;
    CHK(begin(ctx));
    CHK(weights_stage(ctx, Z_dev, N, M, GDCA_MAXQ, -1.0, -1, true, -1));
    CHK(fetch_scalars(ctx));
    *out = ctx->sc_host->pair_sum;
    return GDCA_OK;
}

gdca_status gdca_compute_theta_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, double *theta)
{
    CHK(validate(ctx, N, M, 2));
    if (!Z_dev || !theta) return GDCA_EINVAL;
    CHK(begin(ctx));
    CHK(weights_stage(ctx, Z_dev, N, M, GDCA_MAXQ, -1.0, -1, true, -1));
    CHK(fetch_scalars(ctx));
    *theta = ctx->sc_host->theta;
    return GDCA_OK;
}

gdca_status gdca_neighbour_counts_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, int32_t thresh,
                                      int32_t *n_dev)
{
    CHK(validate(ctx, N, M, 2));
    if (!Z_dev || !n_dev || thresh < 0) return GDCA_EINVAL;
    CHK(begin(ctx));
    CHK(weights_stage(ctx, Z_dev, N, M, GDCA_MAXQ, 0.0, thresh, false, -1));
    HIPCHK(hipMemcpyAsync(n_dev, ctx->nk.p, (size_t)M * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    CHK(fetch_scalars(ctx));
    return symbols_ok(ctx);
}

gdca_status gdca_compute_weights_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, double theta,
                                     double *W_dev, double *Meff, double *theta_used, int32_t *thresh)
{
    CHK(validate(ctx, N, M, 2));
    if (!Z_dev || !W_dev || !Meff || !(theta <= 1.0)) return GDCA_EINVAL;
    CHK(begin(ctx));
    CHK(weights_stage(ctx, Z_dev, N, M, GDCA_MAXQ, theta, -1, false, -1));
    HIPCHK(hipMemcpyAsync(W_dev, ctx->W.p, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    CHK(fetch_scalars(ctx));
    *Meff = ctx->sc_host->Meff;
    if (theta_used) *theta_used = ctx->sc_host->theta;
    if (thresh) *thresh = ctx->sc_host->thresh;
    return symbols_ok(ctx);
}

gdca_status gdca_frequencies_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, int32_t q,
                                 const double *W_dev, double Meff, double *Pi_dev, double *Pij_dev)
{
    CHK(validate(ctx, N, M, q));
    if (!Z_dev || !W_dev || !Pi_dev || !Pij_dev || !(Meff > 0.0)) return GDCA_EINVAL;
    CHK(begin(ctx));
    hipStream_t s = ctx->stream;
    const int sdim = q - 1, n = N * sdim;
    gdca_dev_scalars *sc = (gdca_dev_scalars *)ctx->sc.p;
    CHK(ensure(ctx, ctx->Wfix, (size_t)M * sizeof(unsigned long long)));
    // Meff travels through the device scalar block (the tally kernels read it from HBM)
    HIPCHK(hipMemcpyAsync(&sc->Meff, &Meff, sizeof(double), hipMemcpyHostToDevice, s));
    // W outside [0, 1] would overflow the 59-bit fixed-point weight field: flagged on the device
    gdca_launch_fix_weights(s, W_dev, M, gdca_fix_shift(M), (unsigned long long *)ctx->Wfix.p, sc);
    CHK(tally_stage(ctx, Z_dev, N, M, q, &sc->Meff, 0.0, 0, Pi_dev, Pij_dev, (size_t)n));
    CHK(fetch_scalars(ctx));
    if (ctx->sc_host->bad_symbol & 2) return fail(ctx, GDCA_EINVAL, "weights must lie in [0, 1]%s%s", "", "");
    return symbols_ok(ctx);
}

gdca_status gdca_add_pseudocount_dev(gdca_ctx *ctx, const double *Pi_true_dev, const double *Pij_true_dev, int32_t N,
                                     int32_t q, double pc, double *Pi_dev, double *Pij_dev)
{
    CHK(validate(ctx, N, 1, q));
    if (!Pi_true_dev || !Pij_true_dev || !Pi_dev || !Pij_dev || !(pc >= 0.0 && pc <= 1.0)) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    gdca_launch_add_pseudocount(ctx->stream, Pi_true_dev, Pij_true_dev, N, q, pc, Pi_dev, Pij_dev);
    return check_launch(ctx, "add_pseudocount");
}

gdca_status gdca_covariance_dev(gdca_ctx *ctx, const double *Pi_dev, const double *Pij_dev, int32_t n, double *C_dev)
{
    if (!ctx || !Pi_dev || !Pij_dev || !C_dev || n < 1 || n > GDCA_MAX_N) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    gdca_launch_covariance(ctx->stream, Pi_dev, Pij_dev, n, C_dev);
    return check_launch(ctx, "covariance");
}

// operator-level inverse: kappa_1 = ||C||_1 ||X||_1 (the caller's matrix is still in A_dev) and, beyond the threshold, one
// Newton-Schulz step before the result is copied out.  Synchronises (the caller of an operator waits for `info` anyway).
static gdca_status operator_norms_and_refine(gdca_ctx *ctx, const double *A_dev, int n, int n_pad)
{
    if (ctx->tune.refine == 0) return GDCA_OK;
    gdca_dev_scalars *sc = (gdca_dev_scalars *)ctx->sc.p;
    CHK(ensure(ctx, ctx->normws, (size_t)n_pad * sizeof(double)));
    gdca_launch_matrix_norm1(ctx->stream, A_dev, (size_t)n, n, (double *)ctx->normws.p, &sc->mat_norm1);
    CHK(inverse_norm_stage(ctx, n, n_pad));
    CHK(fetch_scalars(ctx));
    if (!wants_refinement(ctx, *ctx->sc_host)) return GDCA_OK;
    const size_t mat = (size_t)n_pad * n_pad * sizeof(double);
    CHK(ensure(ctx, ctx->C2, mat));
    CHK(ensure(ctx, ctx->B0, mat));
    CHK(ensure(ctx, ctx->Rt, mat));
    gdca_launch_copy_in(ctx->stream, A_dev, n, (double *)ctx->C2.p, n_pad);
    gdca_launch_newton_schulz(ctx->stream, (double *)ctx->A.p, (const double *)ctx->C2.p, (double *)ctx->B0.p, (double *)ctx->Rt.p, n_pad,
                              &sc->ns_resid);
    return check_launch(ctx, "newton_schulz");
}

// ... and the Cholesky fallback where the sweep gave up (non-positive pivot) or the step above cannot have converged.  Leaves the
// member's scalars fetched (sc_host) and its stream idle.
static gdca_status operator_fallback(gdca_ctx *ctx, const double *A_dev, int n, int n_pad)
{
    CHK(fetch_scalars(ctx));
    if (!wants_cholesky(ctx, *ctx->sc_host, ctx->sc_host->ns_resid >= 1.0 ? -1 : 0)) return GDCA_OK;
    CHK(ensure(ctx, ctx->C2, (size_t)n_pad * n_pad * sizeof(double)));
    gdca_launch_copy_in(ctx->stream, A_dev, n, (double *)ctx->C2.p, n_pad);
    CHK(cholesky_stage(ctx, n, n_pad));
    return fetch_scalars(ctx);
}

// the operator-level inverse of the caller's matrix (still intact: the sweep works on a copy), start to finish on the context's stream;
// the context's scalars are in sc_host afterwards
static gdca_status operator_inverse_once(gdca_ctx *ctx, double *A_dev, int n, int n_pad)
{
    hipStream_t s = ctx->stream;
    CHK(ensure(ctx, ctx->A, (size_t)n_pad * n_pad * sizeof(double)));
    gdca_launch_copy_in(s, A_dev, n, (double *)ctx->A.p, n_pad);
    int n_upd = 0;
    CHK(inverse_stage(ctx, n, n_pad, false, &n_upd, nullptr));
    CHK(operator_norms_and_refine(ctx, A_dev, n, n_pad));
    CHK(operator_fallback(ctx, A_dev, n, n_pad));
    if (ctx->sc_host->info == INT32_MIN) return GDCA_OK;  // (the watchdog: the caller's matrix stays as it is)
    gdca_launch_copy_out_neg_sym(s, (const double *)ctx->A.p, n_pad, A_dev, n);
    CHK(check_launch(ctx, "copy_out"));
    HIPCHK(hipStreamSynchronize(s));
    return GDCA_OK;
}

// ... once more after a launch the watchdog ended (gdca_run_collect has the why), up to option SWEEP_RETRIES times
static gdca_status operator_inverse_retry(gdca_ctx *ctx, double *A_dev, int n, int n_pad)
{
    while (ctx->sc_host->info == INT32_MIN && ctx->pend_attempt < ctx->tune.sweep_retries) {
        ++ctx->pend_attempt;
        HIPCHK(hipMemsetAsync(&((gdca_dev_scalars *)ctx->sc.p)->info, 0, sizeof(int), ctx->stream));
        CHK(operator_inverse_once(ctx, A_dev, n, n_pad));
    }
    return GDCA_OK;
}

gdca_status gdca_spd_inverse_dev(gdca_ctx *ctx, double *A_dev, int32_t n, int32_t *info)
{
    if (!ctx || !A_dev || n < 1 || n > GDCA_MAX_N) return GDCA_EINVAL;
    CHK(begin(ctx));
    const int n_pad = round_up(n, GDCA_TILE);
    CHK(operator_inverse_once(ctx, A_dev, n, n_pad));
    CHK(operator_inverse_retry(ctx, A_dev, n, n_pad));
    if (ctx->sc_host->info == INT32_MIN) {
        if (info) *info = 0;
        return fail(ctx, GDCA_EHIP, "SPD inverse aborted: a dependency wait inside the sweep kernel timed out%s%s", "", "");
    }
    if (info) *info = ctx->sc_host->info;
    if (ctx->sc_host->info != 0)
        return fail(ctx, GDCA_ENOTPD, "matrix is not positive definite; Cholesky factorization failed%s%s", "", "");
    return GDCA_OK;
}

gdca_status gdca_spd_inverse_batch_dev(gdca_ctx *const *ctxs, int32_t K, double *const *A_dev, const int32_t *n, int32_t *info)
{
    if (!ctxs || K < 1 || K > 64 || !A_dev || !n) return GDCA_EINVAL;
    gdca_ctx *lead = ctxs[0];
    if (!lead) return GDCA_EINVAL;
    gdca_ctx *ctx = lead;
    for (int k = 0; k < K; ++k) {
        if (!ctxs[k] || !A_dev[k] || n[k] < 1 || n[k] > GDCA_MAX_N) return fail(lead, GDCA_EINVAL, "invalid member of the batch%s%s", "", "");
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return fail(lead, GDCA_EINVAL, "the same context twice in one batch%s%s", "", "");
        if (ctxs[k]->device != lead->device) return fail(lead, GDCA_EINVAL, "contexts of one batch must share a device%s%s", "", "");
        CHK(not_pending(ctxs[k]));
        if (info) info[k] = 0;
    }
    HIPCHK(hipSetDevice(lead->device));
    hipStream_t own[64];
    for (int k = 0; k < K; ++k) {
        own[k] = ctxs[k]->stream;
        if (k > 0) (void)hipStreamSynchronize(own[k]);
        ctxs[k]->stream = lead->stream;
    }
    gdca_status st = GDCA_OK;
    for (int k = 0; k < K && st == GDCA_OK; ++k) {
        gdca_ctx *m = ctxs[k];
        const int n_pad = round_up(n[k], GDCA_TILE);
        st = begin(m);
        if (st == GDCA_OK) st = ensure(m, m->A, (size_t)n_pad * n_pad * sizeof(double));
        if (st != GDCA_OK) {
            member_error(lead, m, k, st);
            break;
        }
        gdca_launch_copy_in(lead->stream, A_dev[k], n[k], (double *)m->A.p, n_pad);
        m->pend_n = n[k];
        m->pend_npad = n_pad;
        m->pend_timed = false;
    }
    if (st == GDCA_OK) st = run_inverses(lead, ctxs, K);
    // kappa_1 and, where it is beyond the threshold, the Newton-Schulz step: member by member, as gdca_spd_inverse_dev does (the
    // caller's matrices are still intact; each member's switches are its own context's)
    for (int k = 0; k < K && st == GDCA_OK; ++k) st = member_error(lead, ctxs[k], k, operator_norms_and_refine(ctxs[k], A_dev[k], n[k], ctxs[k]->pend_npad));
    for (int k = 0; k < K && st == GDCA_OK; ++k) st = member_error(lead, ctxs[k], k, operator_fallback(ctxs[k], A_dev[k], n[k], ctxs[k]->pend_npad));
    for (int k = 0; k < K && st == GDCA_OK; ++k) {
        if (ctxs[k]->sc_host->info == INT32_MIN) continue;  // (the watchdog ended this member's launch: see below)
        gdca_launch_copy_out_neg_sym(lead->stream, (const double *)ctxs[k]->A.p, ctxs[k]->pend_npad, A_dev[k], n[k]);
        st = check_launch(lead, "copy_out");
    }
    (void)hipStreamSynchronize(lead->stream);
    for (int k = 0; k < K; ++k) ctxs[k]->stream = own[k];
    if (st != GDCA_OK) return st;
    // members whose launch the watchdog ended (all members of a merged launch share that fate): once more, one by one
    for (int k = 0; k < K && st == GDCA_OK; ++k)
        if (ctxs[k]->sc_host->info == INT32_MIN) st = member_error(lead, ctxs[k], k, operator_inverse_retry(ctxs[k], A_dev[k], n[k], ctxs[k]->pend_npad));
    if (st != GDCA_OK) return st;
    gdca_status worst = GDCA_OK;
    for (int k = 0; k < K; ++k) {
        gdca_ctx *m = ctxs[k];
        const int inf = m->sc_host->info;   // (fetched by operator_fallback)
        if (inf == INT32_MIN)
            worst = fail(lead, GDCA_EHIP, "SPD inverse aborted: a dependency wait inside the sweep kernel timed out%s%s", "", "");
        else if (inf != 0) {
            if (info) info[k] = inf;
            if (worst == GDCA_OK) worst = fail(lead, GDCA_ENOTPD, "matrix is not positive definite; Cholesky factorization failed%s%s", "", "");
        }
    }
    return worst;
}

// mJ (device, n x n full, ld n) -> ctx->A as "-mJ" with ld = n_pad (the score kernels read its lower triangle)
static gdca_status stage_neg_mJ(gdca_ctx *ctx, const double *mJ_dev, int n, int n_pad)
{
    CHK(ensure(ctx, ctx->A, (size_t)n_pad * n_pad * sizeof(double)));
    gdca_launch_copy_in_neg(ctx->stream, mJ_dev, n, (double *)ctx->A.p, n_pad);
    return check_launch(ctx, "stage_mJ");
}

gdca_status gdca_fn_dev(gdca_ctx *ctx, const double *mJ_dev, int32_t N, int32_t q, double *S_dev)
{
    CHK(validate(ctx, N, 1, q));
    if (!mJ_dev || !S_dev) return GDCA_EINVAL;
    CHK(not_pending(ctx));
    HIPCHK(hipSetDevice(ctx->device));
    const int sdim = q - 1, n = N * sdim, n_pad = round_up(n, GDCA_TILE);
    CHK(stage_neg_mJ(ctx, mJ_dev, n, n_pad));
    return score_stage(ctx, N, sdim, n_pad, GDCA_SCORE_FROB, 0, S_dev);
}

gdca_status gdca_di_dev(gdca_ctx *ctx, const double *mJ_dev, const double *C_dev, int32_t N, int32_t q, double *S_dev)
{
    CHK(validate(ctx, N, 1, q));
    if (!mJ_dev || !C_dev || !S_dev) return GDCA_EINVAL;
    CHK(begin(ctx));
    hipStream_t s = ctx->stream;
    const int sdim = q - 1, n = N * sdim, n_pad = round_up(n, GDCA_TILE);
    CHK(ensure(ctx, ctx->Dblk, (size_t)N * sdim * sdim * sizeof(double)));
    CHK(ensure(ctx, ctx->Ld, (size_t)N * sdim * sdim * sizeof(double)));
    gdca_launch_save_diag_blocks(s, C_dev, (size_t)n, N, sdim, (double *)ctx->Dblk.p);
    gdca_launch_diag_chol(s, (const double *)ctx->Dblk.p, N, sdim, (double *)ctx->Ld.p);
    CHK(stage_neg_mJ(ctx, mJ_dev, n, n_pad));
    CHK(score_stage(ctx, N, sdim, n_pad, GDCA_SCORE_DI, 0, S_dev));
    CHK(fetch_scalars(ctx));
    if (ctx->sc_host->di_noconv)
        return fail(ctx, GDCA_ENOCONV, "eigenvalue iteration of a DI block did not converge%s%s", "", "");
    return GDCA_OK;
}

gdca_status gdca_apc_dev(gdca_ctx *ctx, double *S_dev, int32_t N)
{
    if (!ctx || !S_dev || N < 1) return GDCA_EINVAL;
    CHK(not_pending(ctx));
    HIPCHK(hipSetDevice(ctx->device));
    CHK(ensure(ctx, ctx->colsum, (size_t)N * sizeof(double)));
    gdca_launch_apc(ctx->stream, S_dev, N, (double *)ctx->colsum.p);
    return check_launch(ctx, "apc");
}

// ---- operator level, host pointers: copy in, run the `_dev` form, copy out --------------------------------------
// (what a statement-by-statement caller with ordinary Julia arrays binds; every n x n argument crosses PCIe, so a
// caller who chains several operators should keep the matrices in gdca_dbuf buffers and use the `_dev` forms)

static gdca_status to_dev(gdca_ctx *ctx, gdca_buf &b, const void *host, size_t bytes)
{
    CHK(ensure(ctx, b, bytes));
    HIPCHK(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GDCA_OK;
}

static gdca_status to_host(gdca_ctx *ctx, void *host, const gdca_buf &b, size_t bytes)
{
    HIPCHK(hipMemcpyAsync(host, b.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GDCA_OK;
}

gdca_status gdca_pair_identity_sum(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, uint64_t *out)
{
    CHK(validate(ctx, N, M, 2));
    if (!Z || !out) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    CHK(to_dev(ctx, ctx->scratch[0], Z, (size_t)N * M));
    return gdca_pair_identity_sum_dev(ctx, (const int8_t *)ctx->scratch[0].p, N, M, out);
}

gdca_status gdca_compute_theta(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, double *theta)
{
    CHK(validate(ctx, N, M, 2));
    if (!Z || !theta) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    CHK(to_dev(ctx, ctx->scratch[0], Z, (size_t)N * M));
    return gdca_compute_theta_dev(ctx, (const int8_t *)ctx->scratch[0].p, N, M, theta);
}

gdca_status gdca_neighbour_counts(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, int32_t thresh,
                                  int32_t *n_out)
{
    CHK(validate(ctx, N, M, 2));
    if (!Z || !n_out || thresh < 0) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    CHK(to_dev(ctx, ctx->scratch[0], Z, (size_t)N * M));
    CHK(ensure(ctx, ctx->scratch[2], (size_t)M * sizeof(int32_t)));
    CHK(gdca_neighbour_counts_dev(ctx, (const int8_t *)ctx->scratch[0].p, N, M, thresh, (int32_t *)ctx->scratch[2].p));
    return to_host(ctx, n_out, ctx->scratch[2], (size_t)M * sizeof(int32_t));
}

gdca_status gdca_compute_weights(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, double theta, double *W,
                                 double *Meff, double *theta_used, int32_t *thresh)
{
    CHK(validate(ctx, N, M, 2));
    if (!Z || !W || !Meff || !(theta <= 1.0)) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    CHK(to_dev(ctx, ctx->scratch[0], Z, (size_t)N * M));
    CHK(ensure(ctx, ctx->scratch[2], (size_t)M * sizeof(double)));
    CHK(gdca_compute_weights_dev(ctx, (const int8_t *)ctx->scratch[0].p, N, M, theta, (double *)ctx->scratch[2].p, Meff,
                                 theta_used, thresh));
    return to_host(ctx, W, ctx->scratch[2], (size_t)M * sizeof(double));
}

gdca_status gdca_frequencies(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, int32_t q, const double *W,
                             double Meff, double *Pi, double *Pij)
{
    CHK(validate(ctx, N, M, q));
    if (!Z || !W || !Pi || !Pij || !(Meff > 0.0)) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)N * (q - 1);
    CHK(to_dev(ctx, ctx->scratch[0], Z, (size_t)N * M));
    CHK(to_dev(ctx, ctx->W, W, (size_t)M * sizeof(double)));
    CHK(ensure(ctx, ctx->scratch[1], n * n * sizeof(double)));
    CHK(ensure(ctx, ctx->scratch[2], n * sizeof(double)));
    CHK(gdca_frequencies_dev(ctx, (const int8_t *)ctx->scratch[0].p, N, M, q, (const double *)ctx->W.p, Meff,
                             (double *)ctx->scratch[2].p, (double *)ctx->scratch[1].p));
    CHK(to_host(ctx, Pi, ctx->scratch[2], n * sizeof(double)));
    return to_host(ctx, Pij, ctx->scratch[1], n * n * sizeof(double));
}

gdca_status gdca_add_pseudocount(gdca_ctx *ctx, const double *Pi_true, const double *Pij_true, int32_t N, int32_t q,
                                 double pc, double *Pi, double *Pij)
{
    CHK(validate(ctx, N, 1, q));
    if (!Pi_true || !Pij_true || !Pi || !Pij || !(pc >= 0.0 && pc <= 1.0)) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)N * (q - 1);
    CHK(to_dev(ctx, ctx->scratch[1], Pij_true, n * n * sizeof(double)));
    CHK(to_dev(ctx, ctx->scratch[2], Pi_true, n * sizeof(double)));
    // elementwise: in place on the device
    CHK(gdca_add_pseudocount_dev(ctx, (const double *)ctx->scratch[2].p, (const double *)ctx->scratch[1].p, N, q, pc,
                                 (double *)ctx->scratch[2].p, (double *)ctx->scratch[1].p));
    CHK(to_host(ctx, Pi, ctx->scratch[2], n * sizeof(double)));
    return to_host(ctx, Pij, ctx->scratch[1], n * n * sizeof(double));
}

gdca_status gdca_covariance(gdca_ctx *ctx, const double *Pi, const double *Pij, int32_t n, double *C)
{
    if (!ctx || !Pi || !Pij || !C || n < 1 || n > GDCA_MAX_N) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n;
    CHK(to_dev(ctx, ctx->scratch[1], Pij, nn * nn * sizeof(double)));
    CHK(to_dev(ctx, ctx->scratch[2], Pi, nn * sizeof(double)));
    CHK(gdca_covariance_dev(ctx, (const double *)ctx->scratch[2].p, (const double *)ctx->scratch[1].p, n,
                            (double *)ctx->scratch[1].p));
    return to_host(ctx, C, ctx->scratch[1], nn * nn * sizeof(double));
}

gdca_status gdca_spd_inverse(gdca_ctx *ctx, double *A, int32_t n, int32_t *info)
{
    if (!ctx || !A || n < 1 || n > GDCA_MAX_N) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n;
    CHK(to_dev(ctx, ctx->scratch[1], A, nn * nn * sizeof(double)));
    CHK(gdca_spd_inverse_dev(ctx, (double *)ctx->scratch[1].p, n, info));
    return to_host(ctx, A, ctx->scratch[1], nn * nn * sizeof(double));
}

gdca_status gdca_fn(gdca_ctx *ctx, const double *mJ, int32_t N, int32_t q, double *S)
{
    CHK(validate(ctx, N, 1, q));
    if (!mJ || !S) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)N * (q - 1);
    CHK(to_dev(ctx, ctx->scratch[1], mJ, n * n * sizeof(double)));
    CHK(ensure(ctx, ctx->scratch[2], (size_t)N * N * sizeof(double)));
    CHK(gdca_fn_dev(ctx, (const double *)ctx->scratch[1].p, N, q, (double *)ctx->scratch[2].p));
    return to_host(ctx, S, ctx->scratch[2], (size_t)N * N * sizeof(double));
}

gdca_status gdca_di(gdca_ctx *ctx, const double *mJ, const double *C, int32_t N, int32_t q, double *S)
{
    CHK(validate(ctx, N, 1, q));
    if (!mJ || !C || !S) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)N * (q - 1);
    CHK(to_dev(ctx, ctx->scratch[1], mJ, n * n * sizeof(double)));
    CHK(to_dev(ctx, ctx->scratch[4], C, n * n * sizeof(double)));
    CHK(ensure(ctx, ctx->scratch[2], (size_t)N * N * sizeof(double)));
    CHK(gdca_di_dev(ctx, (const double *)ctx->scratch[1].p, (const double *)ctx->scratch[4].p, N, q,
                    (double *)ctx->scratch[2].p));
    return to_host(ctx, S, ctx->scratch[2], (size_t)N * N * sizeof(double));
}

gdca_status gdca_apc(gdca_ctx *ctx, double *S, int32_t N)
{
    if (!ctx || !S || N < 1) return GDCA_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    CHK(to_dev(ctx, ctx->scratch[2], S, (size_t)N * N * sizeof(double)));
    CHK(gdca_apc_dev(ctx, (double *)ctx->scratch[2].p, N));
    return to_host(ctx, S, ctx->scratch[2], (size_t)N * N * sizeof(double));
}

gdca_status gdca_probe_mfma_f64(gdca_ctx *ctx, int32_t iters, double *tflops)
{
    if (!ctx || !tflops || iters < 1) return GDCA_EINVAL;
    CHK(begin(ctx));
    hipStream_t s = ctx->stream;
    const int blocks = 256 * 2;  // 2 waves per SIMD, all resident at once
    CHK(ensure(ctx, ctx->scratch[6], (size_t)blocks * 256 * sizeof(double)));
    CHK(need_events(ctx, 2));
    gdca_launch_probe_mfma_f64(s, (double *)ctx->scratch[6].p, 16, blocks);  // warm-up
    HIPCHK(hipEventRecord(ctx->ev[0], s));
    gdca_launch_probe_mfma_f64(s, (double *)ctx->scratch[6].p, iters, blocks);
    HIPCHK(hipEventRecord(ctx->ev[1], s));
    CHK(check_launch(ctx, "probe"));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    const double flops = (double)blocks * 4.0 * (double)((iters + 1) / 2) * 16.0 * 2.0 * 16 * 16 * 4;  // 16 MFMAs per trip
    *tflops = flops / ((double)ms * 1e-3) / 1e12;
    return GDCA_OK;
}

}  // extern "C"
