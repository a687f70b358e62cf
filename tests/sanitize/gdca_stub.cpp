// TEST INFRASTRUCTURE, not a fallback: stand-ins for the GPU-side entry points of include/gdca.h that gdca_cli.cpp calls
// (gdca_device_count, gdca_ctx_create / create_peer / destroy, gdca_run, gdca_run_ranked and its _async / _collect halves, gdca_last_error), so that the HOST code of the product --
// gdca_host.cpp (threaded FASTA reader, duplicate removal, ranking sort, writers) and gdca_cli.cpp (parser / worker / writer
// queues of the batch mode) -- can run under AddressSanitizer, UndefinedBehaviorSanitizer and ThreadSanitizer on a machine
// without a GPU (sanitizers on the CPU build only).  Linked ONLY into tests/_build/gdca_cli_{asan,tsan}
// by `make -C gaussdca.jl_amd/csrc asan tsan`; libgdca.so never contains it, and the scores it returns are not gDCA scores:
// a cheap deterministic function of the alignment (fraction of sequences in which two columns agree) that reads every byte
// of Z and writes every entry of S, which is what the sanitizers need.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gdca.h"

struct gdca_ctx {
    int device;
    char err[128];
    // an enqueued ranked run (gdca_run_ranked_async): the stand-in computes at once and keeps S until the collect
    double *pend_S;
    int32_t pend_N, pend_sep;
    gdca_status pend_rc;
    gdca_stats pend_st;
    bool pending;
};

static std::atomic<int> g_live_contexts{0};

extern "C" {

int32_t gdca_device_count(void)
{
    const char *e = getenv("GDCA_STUB_DEVICES");  // how many "GPUs" the batch driver sees
    return e ? atoi(e) : 1;
}

gdca_status gdca_ctx_create(int32_t device_id, gdca_ctx **out)
{
    if (!out) return GDCA_EINVAL;
    *out = nullptr;
    if (device_id < 0 || device_id >= gdca_device_count()) return GDCA_EINVAL;
    if (const char *bad = getenv("GDCA_STUB_FAIL_DEVICE"))  // this device cannot be opened (worker start-up failure path)
        if (atoi(bad) == device_id) return GDCA_EHIP;
    gdca_ctx *c = (gdca_ctx *)calloc(1, sizeof(gdca_ctx));
    if (!c) return GDCA_ENOMEM;
    c->device = device_id;
    ++g_live_contexts;
    *out = c;
    return GDCA_OK;
}

gdca_status gdca_ctx_destroy(gdca_ctx *ctx)
{
    if (!ctx) return GDCA_EINVAL;
    --g_live_contexts;
    free(ctx->pend_S);
    free(ctx);
    return GDCA_OK;
}

const char *gdca_last_error(gdca_ctx *ctx)
{
    return ctx ? ctx->err : "null context";
}

gdca_status gdca_run(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, int32_t q, const gdca_params *p, double *S,
                     gdca_stats *st)
{
    if (!ctx || !Z || !S || !p || N < 1 || M < 1 || q < 2 || q > 31) return GDCA_EINVAL;
    if (p->pseudocount == 0.0) {  // the CLI's PosDefException path
        if (st) {
            memset(st, 0, sizeof(*st));
            st->info = 1;
        }
        snprintf(ctx->err, sizeof ctx->err, "stub: not positive definite");
        return GDCA_ENOTPD;
    }
    for (int32_t i = 0; i < N; ++i)
        for (int32_t j = i; j < N; ++j) {
            long same = 0;
            for (int32_t k = 0; k < M; ++k) {
                const int8_t a = Z[(size_t)k * N + i], b = Z[(size_t)k * N + j];
                if (a < 1 || a > q || b < 1 || b > q) return GDCA_EINVAL;
                same += (a == b);
            }
            const double v = i == j ? 0.0 : (double)same / (double)M + 1e-3 * ((i * 31 + j * 17) % 97);
            S[(size_t)i + (size_t)j * N] = v;
            S[(size_t)j + (size_t)i * N] = v;
        }
    if (st) {
        memset(st, 0, sizeof(*st));
        st->theta = 0.25;
        st->Meff = M;
        st->N = N;
        st->M = M;
        st->q = q;
        st->n = N * (q - 1);
    }
    return GDCA_OK;
}

// (the product sorts on the device; the stand-in uses the host ranking of gdca_host.cpp, which the sanitizer builds link)
gdca_status gdca_run_ranked(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, int32_t q, const gdca_params *p, int32_t min_separation,
                            int32_t *i_out, int32_t *j_out, double *score_out, gdca_stats *st)
{
    if (N < 1 || min_separation < 1) return GDCA_EINVAL;
    double *S = (double *)malloc((size_t)N * N * sizeof(double));
    if (!S) return GDCA_ENOMEM;
    gdca_status rc = gdca_run(ctx, Z, N, M, q, p, S, st);
    if (rc == GDCA_OK) rc = gdca_ranking(S, N, min_separation, i_out, j_out, score_out);
    free(S);
    return rc;
}

gdca_status gdca_ctx_create_peer(gdca_ctx *leader, gdca_ctx **out)
{
    return leader ? gdca_ctx_create(leader->device, out) : GDCA_EINVAL;
}

gdca_status gdca_ctx_set_option(gdca_ctx *ctx, const char *key, const char *value)
{
    return (ctx && key && value) ? GDCA_OK : GDCA_EINVAL;  // (the batch driver sets MERGE_GROUP on its batch contexts)
}

gdca_status gdca_run_ranked_async(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, int32_t q, const gdca_params *p, int32_t min_separation)
{
    if (!ctx || ctx->pending || N < 1 || min_separation < 1) return GDCA_EINVAL;
    free(ctx->pend_S);
    ctx->pend_S = (double *)malloc((size_t)N * N * sizeof(double));
    if (!ctx->pend_S) return GDCA_ENOMEM;
    ctx->pend_rc = gdca_run(ctx, Z, N, M, q, p, ctx->pend_S, &ctx->pend_st);   // (Z is read here and not again: the caller may release it)
    if (ctx->pend_rc == GDCA_EINVAL) return GDCA_EINVAL;                       // argument errors surface at once, as in the product
    ctx->pend_N = N;
    ctx->pend_sep = min_separation;
    ctx->pending = true;
    return GDCA_OK;
}

gdca_status gdca_run_ranked_phased_async(gdca_ctx *const *ctxs, int32_t K, const int8_t *const *Z, const int32_t *N, const int32_t *M,
                                         const int32_t *q, const gdca_params *p, int32_t min_separation)
{
    if (!ctxs || K < 1 || K > 64 || !Z || !N || !M || !q || !p) return GDCA_EINVAL;
    for (int32_t k = 0; k < K; ++k)
        if (!ctxs[k] || ctxs[k]->pending) return GDCA_EINVAL;
    for (int32_t k = 0; k < K; ++k) {
        const gdca_status rc = gdca_run_ranked_async(ctxs[k], Z[k], N[k], M[k], q[k], p, min_separation);
        if (rc != GDCA_OK) {  // (as in the product: nobody is left half-enqueued)
            for (int32_t j = 0; j < k; ++j) ctxs[j]->pending = false;
            return rc;
        }
    }
    return GDCA_OK;
}

gdca_status gdca_run_ranked_collect(gdca_ctx *ctx, int32_t *i_out, int32_t *j_out, double *score_out, gdca_stats *st)
{
    if (!ctx || !ctx->pending) return GDCA_EINVAL;
    ctx->pending = false;
    if (st) *st = ctx->pend_st;
    gdca_status rc = ctx->pend_rc;
    if (rc == GDCA_OK) rc = gdca_ranking(ctx->pend_S, ctx->pend_N, ctx->pend_sep, i_out, j_out, score_out);
    free(ctx->pend_S);
    ctx->pend_S = nullptr;
    return rc;
}

}  // extern "C"
