// Host-side utilities around the hot path (SURVEY.md 8f "next" rows f-1..f-3): the FASTA(.gz) reader
// with gap-fraction filter and letter map (DCAUtils.read_fasta_alignment; reference call site
// src/GaussDCA.jl:20), duplicate removal (DCAUtils.remove_duplicate_sequences; :21-23), the ranking
// enumeration + stable sort (compute_ranking; :88-99) and the "%i %i %e" writer (printrank; :67-74).
// Plain C++ (no HIP): these are host code in the reference too; they live in libgdca.so so that an
// end-to-end gDCA(filename) spends its time on the GPU, not in an interpreter loop.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_set>
#include <vector>

#include <zlib.h>

#include "gdca.h"

struct gdca_fasta {
    int32_t N = 0, M = 0;
    std::vector<int8_t> Z;  // [M][N]
};

namespace {

// ACDEFGHIKLMNPQRSTVWY -> 1..20, everything else -> 21
struct LetterMap {
    int8_t t[256];
    LetterMap()
    {
        for (int i = 0; i < 256; ++i) t[i] = 21;
        const char *L = "ACDEFGHIKLMNPQRSTVWY";
        for (int i = 0; L[i]; ++i) t[(unsigned char)L[i]] = (int8_t)(i + 1);
    }
};
const LetterMap kMap;

bool slurp(const char *path, std::string &out)
{
    gzFile f = gzopen(path, "rb");  // reads plain files transparently
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    std::vector<char> buf(1 << 22);
    for (;;) {
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) {
            gzclose(f);
            return false;
        }
        if (n == 0) break;
        out.append(buf.data(), (size_t)n);
    }
    gzclose(f);
    return true;
}

inline bool is_space(char c)
{
    return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f';
}

}  // namespace

extern "C" {

gdca_status gdca_fasta_open(const char *path, double max_gap_fraction, gdca_fasta **out, int32_t *N, int32_t *M)
{
    if (!path || !out || !N || !M) return GDCA_EINVAL;
    *out = nullptr;
    std::string text;
    if (!slurp(path, text)) return GDCA_EINVAL;

    // records: header line starting with '>', then sequence lines (stripped, concatenated)
    std::vector<std::string> seqs;
    bool in_record = false;
    size_t pos = 0;
    const size_t L = text.size();
    while (pos < L) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = L;
        size_t a = pos, b = eol;
        while (a < b && is_space(text[a])) ++a;
        while (b > a && is_space(text[b - 1])) --b;
        if (b > a) {
            if (text[a] == '>') {
                seqs.emplace_back();
                in_record = true;
            } else if (in_record) {
                seqs.back().append(text, a, b - a);
            }
        }
        pos = eol + 1;
    }
    if (seqs.empty()) return GDCA_EINVAL;

    const std::string &first = seqs[0];
    std::vector<uint32_t> cols;
    for (size_t p = 0; p < first.size(); ++p) {
        const char c = first[p];
        if (c != '.' && !(c >= 'a' && c <= 'z')) cols.push_back((uint32_t)p);
    }
    const int32_t n = (int32_t)cols.size();
    if (n <= 0) return GDCA_EINVAL;

    gdca_fasta *h = new (std::nothrow) gdca_fasta();
    if (!h) return GDCA_ENOMEM;
    h->N = n;
    h->Z.reserve((size_t)n * seqs.size());
    std::vector<int8_t> row((size_t)n);
    for (const std::string &sq : seqs) {
        if (sq.size() != first.size()) {
            delete h;
            return GDCA_EINVAL;  // "inputs are not aligned"
        }
        int ngaps = 0;
        for (int32_t i = 0; i < n; ++i) {
            const unsigned char c = (unsigned char)sq[cols[i]];
            ngaps += (c == '-');
            row[i] = kMap.t[c];
        }
        if ((double)ngaps / (double)n <= max_gap_fraction) {
            h->Z.insert(h->Z.end(), row.begin(), row.end());
            h->M += 1;
        }
    }
    *out = h;
    *N = h->N;
    *M = h->M;
    return GDCA_OK;
}

gdca_status gdca_fasta_copy(const gdca_fasta *h, int8_t *Z)
{
    if (!h || !Z) return GDCA_EINVAL;
    if (!h->Z.empty()) memcpy(Z, h->Z.data(), h->Z.size());
    return GDCA_OK;
}

gdca_status gdca_fasta_close(gdca_fasta *h)
{
    delete h;
    return GDCA_OK;
}

gdca_status gdca_remove_duplicates(const int8_t *Z, int32_t N, int32_t M, int8_t *Z_out, int32_t *keep_idx,
                                   int32_t *M_out)
{
    if (!Z || !Z_out || !M_out || N < 1 || M < 0) return GDCA_EINVAL;
    std::unordered_set<std::string_view> seen;
    seen.reserve((size_t)M * 2);
    int32_t m = 0;
    for (int32_t k = 0; k < M; ++k) {
        const std::string_view key((const char *)Z + (size_t)k * N, (size_t)N);
        if (seen.insert(key).second) {
            if (Z_out + (size_t)m * N != Z + (size_t)k * N) memmove(Z_out + (size_t)m * N, Z + (size_t)k * N, (size_t)N);
            if (keep_idx) keep_idx[m] = k + 1;  // 1-based, as Julia returns them
            ++m;
        }
    }
    *M_out = m;
    return GDCA_OK;
}

int64_t gdca_ranking_length(int32_t N, int32_t min_separation)
{
    if (N < 1 || min_separation < 1 || min_separation >= N) return 0;
    const int64_t d = (int64_t)N - min_separation;
    return d * (d + 1) / 2;
}

gdca_status gdca_ranking(const double *S, int32_t N, int32_t min_separation, int32_t *i_out, int32_t *j_out,
                         double *score_out)
{
    if (!S || N < 1 || min_separation < 1) return GDCA_EINVAL;
    const int64_t len = gdca_ranking_length(N, min_separation);
    if (len == 0) return GDCA_OK;
    if (!i_out || !j_out || !score_out) return GDCA_EINVAL;
    struct Ent {
        int32_t i, j;
        double s;
    };
    std::vector<Ent> R;
    R.reserve((size_t)len);
    for (int32_t i = 1; i <= N - min_separation; ++i)
        for (int32_t j = i + min_separation; j <= N; ++j)
            R.push_back({i, j, S[(size_t)(j - 1) + (size_t)(i - 1) * N]});  // S[j, i], column-major
    // sort!(R, by = x -> x[3], rev = true): stable, so exact ties keep generation order
    std::stable_sort(R.begin(), R.end(), [](const Ent &a, const Ent &b) { return a.s > b.s; });
    for (int64_t t = 0; t < len; ++t) {
        i_out[t] = R[(size_t)t].i;
        j_out[t] = R[(size_t)t].j;
        score_out[t] = R[(size_t)t].s;
    }
    return GDCA_OK;
}

gdca_status gdca_write_rank(const char *path, const int32_t *i, const int32_t *j, const double *score, int64_t len)
{
    if (!path || (len > 0 && (!i || !j || !score))) return GDCA_EINVAL;
    FILE *f = fopen(path, "w");
    if (!f) return GDCA_EINVAL;
    for (int64_t t = 0; t < len; ++t) fprintf(f, "%i %i %e\n", i[t], j[t], score[t]);
    fclose(f);
    return GDCA_OK;
}

// ---- synthetic families ----------------------------------------------------------------------
namespace {
struct SplitMix {
    uint64_t s;
    // stream (tag, idx) of a seed: the start state is itself a SplitMix64 output, so streams do not overlap in practice
    SplitMix(uint64_t seed, uint64_t tag, uint64_t idx) : s(seed ^ (tag << 56) ^ idx) { s = next(); }
    uint64_t next()
    {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};
// one draw per site: high 32 bits against the threshold, low 32 bits pick the replacement symbol 1..q-1
inline int8_t resample(uint64_t r, uint32_t thresh, uint32_t nsym, int8_t keep)
{
    return (uint32_t)(r >> 32) < thresh ? (int8_t)(1 + (((r & 0xFFFFFFFFull) * nsym) >> 32)) : keep;
}
}  // namespace

gdca_status gdca_synth_family(int32_t N, int32_t M, int32_t q, uint64_t seed, int8_t *Z)
{
    if (!Z || N < 1 || M < 1 || q < 2 || q > 31) return GDCA_EINVAL;
    static const uint32_t kMu[6] = {85899345u, 214748364u, 429496729u, 858993459u, 1288490188u, 2147483648u};
    const uint32_t nsym = (uint32_t)(q - 1);
    std::vector<int8_t> root((size_t)N);
    {
        SplitMix g(seed, 0, 0);
        for (int32_t i = 0; i < N; ++i) root[(size_t)i] = (int8_t)(1 + g.below(nsym));
    }
    const int32_t K = (M + 24) / 25;
    std::vector<int8_t> centres((size_t)K * N);
    for (int32_t c = 0; c < K; ++c) {
        SplitMix g(seed, 1, (uint64_t)c);
        for (int32_t i = 0; i < N; ++i) centres[(size_t)c * N + i] = resample(g.next(), 1u << 30, nsym, root[(size_t)i]);
    }
    const uint32_t maxlen = (uint32_t)std::max(2, N / 10);
    for (int32_t k = 0; k < M; ++k) {
        SplitMix g(seed, 2, (uint64_t)k);
        const int8_t *cen = centres.data() + (size_t)g.below((uint32_t)K) * N;
        const uint32_t thr = kMu[g.below(6)];
        int8_t *row = Z + (size_t)k * N;
        for (int32_t i = 0; i < N; ++i) row[i] = resample(g.next(), thr, nsym, cen[i]);
        const uint32_t nruns = g.below(4);
        for (uint32_t r = 0; r < nruns; ++r) {
            const uint32_t a = g.below((uint32_t)N);
            const uint32_t len = 1 + g.below(maxlen);
            for (uint32_t i = a; i < std::min((uint32_t)N, a + len); ++i) row[i] = (int8_t)q;
        }
    }
    return GDCA_OK;
}

gdca_status gdca_write_fasta(const char *path, const int8_t *Z, int32_t N, int32_t M)
{
    if (!path || !Z || N < 1 || M < 0) return GDCA_EINVAL;
    static const char L[] = "?ACDEFGHIKLMNPQRSTVWY-";
    std::string out;
    out.reserve((size_t)M * ((size_t)N + 16));
    char hdr[32];
    for (int32_t k = 0; k < M; ++k) {
        out.append(hdr, (size_t)snprintf(hdr, sizeof hdr, ">s%d\n", k));
        for (int32_t i = 0; i < N; ++i) {
            const int8_t a = Z[(size_t)k * N + i];
            if (a < 1 || a > 21) return GDCA_EINVAL;
            out.push_back(L[a]);
        }
        out.push_back('\n');
    }
    const size_t plen = strlen(path);
    if (plen > 3 && strcmp(path + plen - 3, ".gz") == 0) {
        gzFile f = gzopen(path, "wb1");
        if (!f) return GDCA_EINVAL;
        size_t off = 0;
        while (off < out.size()) {
            const unsigned n = (unsigned)std::min<size_t>(out.size() - off, 1u << 30);
            if (gzwrite(f, out.data() + off, n) != (int)n) {
                gzclose(f);
                return GDCA_EINVAL;
            }
            off += n;
        }
        return gzclose(f) == Z_OK ? GDCA_OK : GDCA_EINVAL;
    }
    FILE *f = fopen(path, "wb");
    if (!f) return GDCA_EINVAL;
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return (fclose(f) == 0 && ok) ? GDCA_OK : GDCA_EINVAL;
}

}  // extern "C"
