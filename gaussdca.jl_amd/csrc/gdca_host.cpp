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

}  // extern "C"
