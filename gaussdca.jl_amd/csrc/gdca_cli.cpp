// gdca_cli -- command-line driver over the C-ABI of libgdca.so (include/gdca.h): what a user of the reference
// types at the Julia prompt (README.md "Examples": gDCA("alignment.fasta.gz", pseudocount = 0.2, score = :DI);
// printrank("results_DI.txt", DIR)), plus the directory-batch mode of SURVEY.md 8e / 8f-4: independent families
// over every GPU of the node, one gdca_ctx + one worker thread per GPU pulling from a queue ordered by
// descending cost (greedy LPT), with FASTA parsing done by separate threads so that it overlaps device work.
// No collective, no shared device state.  Host code only; every number comes from libgdca.so.
//
//   gdca_cli [options] alignment.fasta[.gz] [ranking.txt]
//   gdca_cli [options] --batch DIR --out OUTDIR [--gpus G] [--parsers P] [--inflight K]
//   gdca_cli [options] --batch DIR --parse-only [--parsers P]     (host side alone: read + filter + map, no GPU needed)
//   gdca_cli --synth N M SEED out.fasta[.gz]
// GDCA_VISIBLE_DEVICES=0,2,5 restricts (and orders) the HIP devices the batch mode uses.
// options (names and defaults of src/GaussDCA.jl:10-15):
//   --pseudocount X (0.8)  --theta auto|X (auto)  --max_gap_fraction X (0.9)  --score frob|DI (frob)
//   --min_separation K (5)  --remove_dups
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <malloc.h>
#include <thread>
#include <vector>

#include <dirent.h>
#include <sys/stat.h>

#include "gdca.h"

namespace {

struct Options {
    double pseudocount = 0.8, theta = -1.0, max_gap_fraction = 0.9;
    int score = GDCA_SCORE_FROB, min_separation = 5;
    bool remove_dups = false;
    std::string batch_dir, out_dir;
    bool merge_given = false;  // --merge on the command line (then the library's own pivot grouping; by default single-block groups: the same bits as unmerged runs)
    int gpus = 0, parsers = 0 /* 0 = hardware threads / 8, between 4 and 32 */, inflight = 2, passes = 1, merge = 8 /* families per phase batch (--merge K, up to 32); 1 = off: see the worker */, merge_blocks = 24 /* largest covariance of a "small" family, in 128-blocks */;
    bool parse_only = false;
    std::vector<std::string> positional;
};

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

[[noreturn]] void die(const std::string &msg)
{
    fprintf(stderr, "ERROR: %s\n", msg.c_str());
    exit(2);
}

// the checks of check_arguments (src/GaussDCA.jl:49-65), same order and wording
void check_arguments(const Options &o)
{
    char buf[256];
    if (!(o.pseudocount >= 0 && o.pseudocount <= 1)) {
        snprintf(buf, sizeof buf, "invalid pseudocount value: %g (must be between 0 and 1)", o.pseudocount);
        die(buf);
    }
    if (!(o.theta < 0 || o.theta <= 1)) {
        snprintf(buf, sizeof buf, "invalid theta value: %g (must be either :auto, or a number between 0 and 1)", o.theta);
        die(buf);
    }
    if (!(o.max_gap_fraction >= 0 && o.max_gap_fraction <= 1)) {
        snprintf(buf, sizeof buf, "invalid max_gap_fraction value: %g (must be between 0 and 1)", o.max_gap_fraction);
        die(buf);
    }
    if (!(o.min_separation >= 1)) {
        snprintf(buf, sizeof buf, "invalid min_separation value: %d (must be >= 1)", o.min_separation);
        die(buf);
    }
}

struct Family {
    std::string path, name;
    int32_t N = 0, M = 0, q = 0;
    gdca_fasta *h = nullptr;  // the parsed alignment, used where the reader left it ...
    std::vector<int8_t> Zown; // ... unless duplicate removal rewrote it: [M][N]
    double parse_s = 0;
    std::string error;
    const int8_t *Z() const { return Zown.empty() ? gdca_fasta_data(h) : Zown.data(); }
    Family() = default;
    Family(const Family &) = delete;
    Family &operator=(const Family &) = delete;
    Family(Family &&o) noexcept { *this = std::move(o); }
    Family &operator=(Family &&o) noexcept
    {
        if (this != &o) {
            if (h) gdca_fasta_close(h);
            path = std::move(o.path);
            name = std::move(o.name);
            N = o.N, M = o.M, q = o.q, parse_s = o.parse_s;
            h = o.h;
            o.h = nullptr;
            Zown = std::move(o.Zown);
            error = std::move(o.error);
        }
        return *this;
    }
    ~Family()
    {
        if (h) gdca_fasta_close(h);
    }
};

// src/GaussDCA.jl:20-26: read + filter, optional duplicate removal, q = maximum(Z), q < 32
bool load_family(const Options &o, Family &f)
{
    const double t0 = now();
    if (gdca_fasta_open(f.path.c_str(), o.max_gap_fraction, &f.h, &f.N, &f.M) != GDCA_OK) {
        f.error = "cannot read alignment " + f.path;
        return false;
    }
    int q = gdca_fasta_max_symbol(f.h);  // q = maximum(Z) (src/GaussDCA.jl:25), found by the reader's threads
    if (o.remove_dups) {
        f.Zown.resize((size_t)f.N * std::max(f.M, 1));
        int32_t m = 0;
        if (gdca_remove_duplicates(gdca_fasta_data(f.h), f.N, f.M, f.Zown.data(), nullptr, &m) != GDCA_OK) {
            f.error = "duplicate removal failed for " + f.path;
            return false;
        }
        f.M = m;
        f.Zown.resize((size_t)f.N * std::max(f.M, 1));
        gdca_fasta_close(f.h);
        f.h = nullptr;
        q = 0;
        for (size_t x = 0; x < (size_t)f.N * f.M; ++x) q = std::max(q, (int)f.Zown[x]);
    }
    f.q = q;
    f.parse_s = now() - t0;
    if (f.M < 1) {
        f.error = "no sequences left after filtering in " + f.path;
        return false;
    }
    if (q >= 32) {
        f.error = "parameter q=" + std::to_string(q) + " is too big (max 31 is allowed)";
        return false;
    }
    return true;
}

// the ranking of one family as the library hands it back (src/GaussDCA.jl:28-44: hot path + compute_ranking on the device)
struct RankOut {
    std::vector<int32_t> i, j;
    std::vector<double> s;
};

bool compute(gdca_ctx *ctx, const Options &o, const Family &f, RankOut &R, gdca_stats *st, std::string *err)
{
    const int64_t len = std::max<int64_t>(gdca_ranking_length(f.N, o.min_separation), 0);
    R.i.resize((size_t)len);
    R.j.resize((size_t)len);
    R.s.resize((size_t)len);
    gdca_params p{o.pseudocount, o.theta, o.score, 1};
    const gdca_status rc = gdca_run_ranked(ctx, f.Z(), f.N, f.M, f.q, &p, o.min_separation, R.i.data(), R.j.data(), R.s.data(), st);
    if (rc == GDCA_ENOTPD) {
        *err = "PosDefException: matrix is not positive definite; Cholesky factorization failed (info " +
               std::to_string(st->info) + ")";
        return false;
    }
    if (rc != GDCA_OK) {
        *err = std::string("gdca_run failed: ") + gdca_last_error(ctx);
        return false;
    }
    return true;
}

// printrank (src/GaussDCA.jl:67-74): host work, off the GPU worker's thread in batch mode
bool emit(const RankOut &R, const std::string &out_path, std::string *err)
{
    const int64_t len = (int64_t)R.i.size();
    if (out_path.empty()) {
        for (int64_t t = 0; t < len; ++t) printf("%i %i %e\n", R.i[(size_t)t], R.j[(size_t)t], R.s[(size_t)t]);
    } else if (gdca_write_rank(out_path.c_str(), R.i.data(), R.j.data(), R.s.data(), len) != GDCA_OK) {
        *err = "cannot write " + out_path;
        return false;
    }
    return true;
}

bool process(gdca_ctx *ctx, const Options &o, const Family &f, const std::string &out_path, gdca_stats *st,
             std::string *err)
{
    RankOut R;
    return compute(ctx, o, f, R, st, err) && emit(R, out_path, err);
}

bool has_suffix(const std::string &s, const char *suf)
{
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

std::string strip_fasta_suffix(std::string name)
{
    if (has_suffix(name, ".gz")) name.resize(name.size() - 3);
    for (const char *suf : {".fasta", ".fa", ".afa", ".aln"})
        if (has_suffix(name, suf)) {
            name.resize(name.size() - strlen(suf));
            break;
        }
    return name;
}

// A cheap size estimate for the scheduling order without parsing: compressed/plain bytes on disk.  The exact
// cost model c = (N s)^3 + M^2 N / 8 + N^2 M (batch.py: family_cost) is applied once (N, M) are known; the queue
// order only needs "big first", for which the file size is a good proxy (bytes ~ N M).
struct Job {
    std::string path, name;
    int64_t bytes = 0;
};

// GDCA_VISIBLE_DEVICES: comma-separated HIP device ids the batch mode may use, in that order; unset = all of them.
// Ids outside 0..ndev-1 and repeats are an error (a typo must not silently shrink the job to fewer GPUs).
std::vector<int> visible_devices(int ndev)
{
    std::vector<int> ids;
    const char *env = getenv("GDCA_VISIBLE_DEVICES");
    if (!env || !*env) {
        for (int g = 0; g < ndev; ++g) ids.push_back(g);
        return ids;
    }
    const std::string s = env;
    size_t pos = 0;
    while (pos <= s.size()) {
        const size_t comma = std::min(s.find(',', pos), s.size());
        const std::string tok = s.substr(pos, comma - pos);
        char *end = nullptr;
        const long v = strtol(tok.c_str(), &end, 10);
        if (tok.empty() || *end != 0 || v < 0 || v >= ndev || std::find(ids.begin(), ids.end(), (int)v) != ids.end())
            die("invalid GDCA_VISIBLE_DEVICES entry '" + tok + "' (" + std::to_string(ndev) + " HIP device(s) present)");
        ids.push_back((int)v);
        pos = comma + 1;
    }
    return ids;
}

int run_batch(const Options &o)
{
    std::vector<Job> jobs;
    DIR *d = opendir(o.batch_dir.c_str());
    if (!d) die("cannot open directory " + o.batch_dir);
    while (dirent *e = readdir(d)) {
        const std::string nm = e->d_name;
        bool ok = false;
        for (const char *suf : {".fasta", ".fa", ".afa", ".aln", ".fasta.gz", ".fa.gz", ".afa.gz", ".aln.gz"})
            ok = ok || has_suffix(nm, suf);
        if (!ok) continue;
        Job j;
        j.path = o.batch_dir + "/" + nm;
        j.name = strip_fasta_suffix(nm);
        struct stat sb;
        if (stat(j.path.c_str(), &sb) == 0) j.bytes = has_suffix(nm, ".gz") ? (int64_t)sb.st_size * 6 : (int64_t)sb.st_size;
        jobs.push_back(j);
    }
    closedir(d);
    if (jobs.empty()) die("no FASTA files in " + o.batch_dir);
    // big first; name breaks ties so that the order is a function of the directory contents alone
    std::sort(jobs.begin(), jobs.end(), [](const Job &a, const Job &b) { return a.bytes != b.bytes ? a.bytes > b.bytes : a.name < b.name; });
    // several files are parsed at once: share the host threads between the parser threads instead of letting every
    // gdca_fasta_open start 16 of its own
    // (measured on the 256-thread host of the GPU box, tools/parse_bench.sh: with 8 or more files in flight nested threads only
    // add contention in the kernel's memory-map lock -- 946 families/s with one thread per file against 497 with eight)
    if (!getenv("GDCA_FASTA_THREADS")) {
        const unsigned hw = (unsigned)std::max(1, gdca_host_cpus());
        const unsigned per_file = o.parsers >= 8 ? 1u : std::max(1u, std::min(16u, hw / (unsigned)std::max(1, o.parsers)));
        setenv("GDCA_FASTA_THREADS", std::to_string(per_file).c_str(), 1);
    }
    // dozens of parser threads allocating and freeing megabyte-sized vectors per file: keep those blocks inside malloc's arenas
    // instead of one mmap / munmap pair each (both take the process-wide memory-map lock)
    (void)mallopt(M_MMAP_THRESHOLD, 1 << 30);
    (void)mallopt(M_TRIM_THRESHOLD, 1 << 30);
    if (o.parse_only) {
        // the host side of the batch alone: P parser threads over the whole directory (read, inflate, column filter, letter
        // map, gap filter, optional duplicate removal), results dropped.  This is the rate the GPUs of a node have to be fed
        // at: 8 GPUs x ~70 families/s need it to be several hundred families per second.
        // --passes R: the directory R times over by the SAME threads, passes 2 .. R timed as one stream of (R - 1) x files with no
        // barrier between passes -- the steady state of a long batch: the threads' text buffers and the pooled matrices exist and
        // are faulted in, and the one 48 MB family that a single thread needs ~0.15 s for no longer IS the measurement (a pass
        // over a few hundred files takes about that long)
        std::atomic<size_t> fails{0};
        std::atomic<long long> seqs{0}, cells{0};
        const int P = std::max(1, o.parsers), R = std::max(1, o.passes);
        std::vector<std::atomic<size_t>> next((size_t)R);
        for (auto &x : next) x = 0;
        std::atomic<int> arrived{0};
        std::atomic<double> t_last{0.0};
        std::vector<std::thread> th;
        for (int p = 0; p < P; ++p)
            th.emplace_back([&] {
                for (int r = 0; r < R; ++r) {
                    if (r == (R > 1 ? 1 : 0) && arrived++ == 0) t_last = now();  // the first thread to leave the warm-up pass starts the clock
                    for (size_t idx; (idx = next[(size_t)r]++) < jobs.size();) {
                        Family f;
                        f.path = jobs[idx].path;
                        f.name = jobs[idx].name;
                        if (!load_family(o, f)) {
                            if (r == R - 1) {
                                fprintf(stderr, "ERROR: %s\n", f.error.c_str());
                                ++fails;
                            }
                            continue;
                        }
                        if (r >= (R > 1 ? 1 : 0)) {
                            seqs += f.M;
                            cells += (long long)f.M * f.N;
                        }
                    }
                }
            });
        for (auto &t : th) t.join();
        const double wall = now() - t_last.load();
        long long bytes = 0;
        for (const Job &j : jobs) {
            struct stat sb;
            if (stat(j.path.c_str(), &sb) == 0) bytes += (long long)sb.st_size;
        }
        const int timed = R > 1 ? R - 1 : 1;
        fprintf(stderr, "parse-only: %zu families x %d timed pass(es) (%lld sequences, %.1f MB on disk per pass, %.1f M symbols kept) on %d parser "
                        "thread(s) in %.3f s = %.1f families/s, %.1f MB/s (%zu failed)\n",
                jobs.size(), timed, seqs.load(), bytes / 1e6, cells.load() / 1e6, std::max(1, o.parsers), wall, jobs.size() * (double)timed / wall,
                bytes / 1e6 * timed / wall, fails.load());
        return fails.load() ? 1 : 0;
    }
    mkdir(o.out_dir.c_str(), 0777);

    const int ndev = gdca_device_count();
    if (ndev < 1) die("no HIP device (there is no CPU fallback)");
    std::vector<int> devs = visible_devices(ndev);
    if (o.gpus > 0 && (size_t)o.gpus < devs.size()) devs.resize((size_t)o.gpus);
    const int G = (int)devs.size();

    // parser threads fill a bounded queue of parsed families in job order; GPU workers pull from it
    std::mutex mu;
    std::condition_variable cv_ready, cv_space;
    std::deque<Family> ready;
    size_t next_job = 0, parsed_done = 0;
    const size_t cap = (size_t)std::max(2 * G * std::max(1, o.inflight), 4);
    std::atomic<int> failures{0};
    // GPU workers still able to take families; when the last one is gone (every gdca_ctx_create failed) the parsers
    // are told to stop: nobody would ever drain `ready`, and they would wait on cv_space for ever
    int live_workers = 0;
    bool abort_parsers = false;

    auto parser = [&]() {
        for (;;) {
            size_t idx;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_space.wait(lk, [&] { return abort_parsers || ready.size() < cap || next_job >= jobs.size(); });
                if (abort_parsers || next_job >= jobs.size()) return;
                idx = next_job++;
                if (next_job >= jobs.size()) cv_space.notify_all();
            }
            Family f;
            f.path = jobs[idx].path;
            f.name = jobs[idx].name;
            load_family(o, f);  // errors travel with the family and are reported by the worker
            {
                std::lock_guard<std::mutex> lk(mu);
                ready.push_back(std::move(f));
                ++parsed_done;
            }
            cv_ready.notify_all();
        }
    };
    // rankings (sorted on the device) are formatted and written by their own threads: the GPU worker goes straight to the next family
    struct Result {
        std::string name;
        RankOut R;
    };
    std::mutex omu;
    std::condition_variable cv_out;
    std::deque<Result> outq;
    bool workers_done = false;
    auto writer = [&]() {
        for (;;) {
            Result r;
            {
                std::unique_lock<std::mutex> lk(omu);
                cv_out.wait(lk, [&] { return !outq.empty() || workers_done; });
                if (outq.empty()) return;
                r = std::move(outq.front());
                outq.pop_front();
            }
            std::string err;
            if (!emit(r.R, o.out_dir + "/" + r.name + ".rank.txt", &err)) {
                fprintf(stderr, "ERROR: %s: %s\n", r.name.c_str(), err.c_str());
                ++failures;
            }
        }
    };
    const double t0 = now();
    std::vector<double> busy((size_t)G, 0.0);
    std::vector<int> count((size_t)G, 0);
    std::atomic<int> merged_batches{0};
    std::vector<double> done_at;  // completion time of every family, in order (steady-state rate: start-up -- process start, HIP initialisation,
                                  // the first parses, the first batch -- excluded)
    int completed = 0;
    // One worker thread per GPU drives a PIPELINE of --inflight contexts (the leader and its peers, gdca_ctx_create_peer): family k+1
    // is uploaded and enqueued while family k computes, and family k is collected only when its context is needed again -- the GPU
    // never waits for an upload, a download or the host's launch latencies.  (Round 3 ran --inflight independent workers with
    // synchronous calls instead: their kernels interleaved without any order.)
    auto worker = [&](int g) {  // g = slot in `devs`
        struct Slot {
            gdca_ctx *ctx = nullptr;
            bool busy = false;
            Result res;
            int32_t N = 0, M = 0, q = 0;
            double parse_s = 0.0, t_start = 0.0;
        };
        const int K = std::max(1, o.inflight);
        std::vector<Slot> slots((size_t)K);
        bool created = gdca_ctx_create(devs[(size_t)g], &slots[0].ctx) == GDCA_OK;
        for (int k = 1; k < K && created; ++k) created = gdca_ctx_create_peer(slots[0].ctx, &slots[(size_t)k].ctx) == GDCA_OK;
        if (!created) {
            fprintf(stderr, "ERROR: cannot create a context on GPU %d\n", devs[(size_t)g]);
            ++failures;
            for (int k = K - 1; k >= 0; --k)
                if (slots[(size_t)k].ctx) gdca_ctx_destroy(slots[(size_t)k].ctx);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--live_workers == 0) abort_parsers = true;
            }
            cv_space.notify_all();
            cv_ready.notify_all();
            return;
        }
        double busy_s = 0.0;
        int done = 0;
        auto finish = [&](Slot &sl) {
            const double t = now();
            gdca_stats st{};
            const gdca_status rc = gdca_run_ranked_collect(sl.ctx, sl.res.R.i.data(), sl.res.R.j.data(), sl.res.R.s.data(), &st);
            sl.busy = false;
            busy_s += now() - t;
            ++done;
            {
                std::lock_guard<std::mutex> lk(omu);
                done_at.push_back(now());
                ++completed;
            }
            if (rc != GDCA_OK) {
                if (rc == GDCA_ENOTPD)
                    fprintf(stderr, "ERROR: %s: PosDefException: matrix is not positive definite; Cholesky factorization failed (info %d)\n",
                            sl.res.name.c_str(), st.info);
                else
                    fprintf(stderr, "ERROR: %s: gdca_run failed: %s\n", sl.res.name.c_str(), gdca_last_error(sl.ctx));
                ++failures;
                return;
            }
            char again[96] = "";
            if (st.sweep_retries > 0) snprintf(again, sizeof(again), "  (SPD inverse run again, %d time(s): the watchdog had ended its launch)", st.sweep_retries);
            fprintf(stderr, "gpu %d  %-24s N=%d M=%d q=%d theta=%.6f Meff=%.4f  parse %.3fs  device %.1f ms  total %.3fs%s\n", devs[(size_t)g],
                    sl.res.name.c_str(), sl.N, sl.M, sl.q, st.theta, st.Meff, sl.parse_s, st.ms_total, now() - sl.t_start, again);
            {
                std::lock_guard<std::mutex> lk(omu);
                outq.push_back(std::move(sl.res));
            }
            cv_out.notify_one();
        };
        // Small families (covariance of at most --merge-blocks 128-blocks: chain-bound inverses that leave most of the chip idle) do
        // not go through the slots one by one but up to --merge at a time through gdca_run_ranked_phased_async: one phase-batched
        // run whose SPD inverses share launches of the sweep kernel.  Two sets of contexts take turns, so that the next batch is
        // uploaded and enqueued while the previous one computes; they are created when the first small family shows up -- as peers
        // of the pipeline's leader: ONE gate orders every SPD inverse of this GPU, single or merged (two persistent sweep launches
        // that become resident side by side can wait for each other's workgroups until the watchdog ends them).
        // ON by default for families of up to 24 blocks (N <= 153 at q = 21), eight to a batch, with the members' pivot groups those of
        // single launches (MERGE_GROUP=1: every ranking file byte for byte what `--merge 1` writes; tests/test_gpu_configs.py).  `--merge K`
        // on the command line selects the library's own grouping of merged members (larger pivot groups: scores equal to rounding,
        // 1e-13, and ~7 % faster at config B's size), `--merge-blocks B` which families count as small, `--merge 1` switches it off.
        // Measured on one MI355X (profiles/r06_cli_batch.log, r05_cli_merge.log): 96 families of config B's size 661-665 families/s through
        // the slots' pipeline, 1231-1243 through merged batches of eight; on the mixed batch of configuration E batches of its small
        // families neither gain nor lose (round 5: 72.4-72.8 against 72.8), and phase batches of EVERYTHING (--merge 16 --merge-blocks
        // 100000: all 256 families, files to files, 3.73-3.81 s against 3.66-3.75 s) lose at the process level what they win on the
        // device: the first batch waits for sixteen parses, and the worker uploads and collects sixteen families at a time.
        const int SMALL_BLOCKS = o.merge_blocks;
        struct Set {
            std::vector<Slot> mem;
            int used = 0;
        };
        Set sets[2];
        int cur_set = 0;
        bool sets_ok = true;
        // (by default only in directories of 128 files and more: the two sets' sixteen extra contexts cost ~40 ms once, a fifth of a
        // 96-file run of config B's size -- 0.25-0.28 s against 0.22 s -- and nothing of a long one)
        const char *min_files = getenv("GDCA_CLI_MERGE_MIN_FILES");  // (tests: the default batches in a small directory)
        const bool merging = o.merge > 1 && (o.merge_given || jobs.size() >= (size_t)(min_files ? std::max(1, atoi(min_files)) : 128));
        // the sets' contexts are made NOW, before anything runs on the GPU: a context's stream brings a hardware queue to life, and
        // the driver takes all running kernels off the device and back for that (gdca_api.hip, warm_stream)
        if (merging && !getenv("GDCA_CLI_LAZY_SETS"))
            for (Set &st : sets)
                while (sets_ok && (int)st.mem.size() < o.merge) {
                    Slot sl;
                    if (gdca_ctx_create_peer(slots[0].ctx, &sl.ctx) != GDCA_OK) sets_ok = false;
                    else {
                        if (!o.merge_given && !getenv("GDCA_MERGE_GROUP")) (void)gdca_ctx_set_option(sl.ctx, "MERGE_GROUP", "1");
                        st.mem.push_back(std::move(sl));
                    }
                }
        auto is_small = [&](const Family &f) { return merging && ((long long)f.N * (f.q - 1) + 127) / 128 <= SMALL_BLOCKS; };
        auto finish_set = [&](Set &st) {
            for (int m = 0; m < st.used; ++m)
                if (st.mem[(size_t)m].busy) finish(st.mem[(size_t)m]);
            st.used = 0;
        };
        auto drain = [&](size_t k) {
            for (int j = 0; j < K; ++j) {
                Slot &t = slots[(k + (size_t)j) % (size_t)K];
                if (t.busy) finish(t);
            }
            finish_set(sets[cur_set]);
            finish_set(sets[cur_set ^ 1]);
        };
        auto fill_slot = [&](Slot &sl, const Family &f, double t) {
            const int64_t len = std::max<int64_t>(gdca_ranking_length(f.N, o.min_separation), 0);
            sl.res = Result();
            sl.res.name = f.name;
            sl.res.R.i.resize((size_t)len);
            sl.res.R.j.resize((size_t)len);
            sl.res.R.s.resize((size_t)len);
            sl.N = f.N;
            sl.M = f.M;
            sl.q = f.q;
            sl.parse_s = f.parse_s;
            sl.t_start = t;
        };
        for (size_t k = 0;; ++k) {
            Family f;
            bool got = false;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (!ready.empty()) {
                    f = std::move(ready.front());
                    ready.pop_front();
                    got = true;
                }
            }
            if (!got) {
                // nothing parsed right now: hand over what has been computed before waiting for the parsers
                drain(k);
                std::unique_lock<std::mutex> lk(mu);
                cv_ready.wait(lk, [&] { return !ready.empty() || parsed_done >= jobs.size(); });
                if (ready.empty()) break;
                f = std::move(ready.front());
                ready.pop_front();
            }
            cv_space.notify_one();
            if (!f.error.empty()) {
                fprintf(stderr, "ERROR: %s\n", f.error.c_str());
                ++failures;
                --k;  // the slot stays free
                continue;
            }
            if (sets_ok && is_small(f)) {
                // a batch: this family and whatever small ones the parsers have ready behind it
                std::vector<Family> grp;
                grp.push_back(std::move(f));
                {
                    std::unique_lock<std::mutex> lk(mu);
                    while ((int)grp.size() < o.merge && !ready.empty() && ready.front().error.empty() && is_small(ready.front())) {
                        grp.push_back(std::move(ready.front()));
                        ready.pop_front();
                    }
                }
                cv_space.notify_all();
                Set &st = sets[cur_set];
                finish_set(st);  // (its previous batch; the other set's keeps computing meanwhile)
                const double t = now();
                while (sets_ok && st.mem.size() < grp.size()) {
                    Slot sl;
                    if (gdca_ctx_create_peer(slots[0].ctx, &sl.ctx) != GDCA_OK) sets_ok = false;
                    else {
                        // (the default batches keep the pivot grouping of single launches: every ranking file is byte for byte what the
                        // unmerged driver writes; --merge K asks for the library's faster grouping of merged members, equal to rounding)
                        if (!o.merge_given && !getenv("GDCA_MERGE_GROUP")) (void)gdca_ctx_set_option(sl.ctx, "MERGE_GROUP", "1");
                        st.mem.push_back(std::move(sl));
                    }
                }
                bool started = false;
                if (sets_ok) {
                    const int G2 = (int)grp.size();
                    std::vector<gdca_ctx *> cs((size_t)G2);
                    std::vector<const int8_t *> zs((size_t)G2);
                    std::vector<int32_t> ns((size_t)G2), ms((size_t)G2), qs((size_t)G2);
                    for (int m = 0; m < G2; ++m) {
                        fill_slot(st.mem[(size_t)m], grp[(size_t)m], t);
                        cs[(size_t)m] = st.mem[(size_t)m].ctx;
                        zs[(size_t)m] = grp[(size_t)m].Z();
                        ns[(size_t)m] = grp[(size_t)m].N;
                        ms[(size_t)m] = grp[(size_t)m].M;
                        qs[(size_t)m] = grp[(size_t)m].q;
                    }
                    gdca_params p{o.pseudocount, o.theta, o.score, 1};
                    const gdca_status rc = gdca_run_ranked_phased_async(cs.data(), G2, zs.data(), ns.data(), ms.data(), qs.data(), &p, o.min_separation);
                    busy_s += now() - t;
                    if (rc == GDCA_OK) {
                        for (int m = 0; m < G2; ++m) st.mem[(size_t)m].busy = true;
                        st.used = G2;
                        ++merged_batches;
                        cur_set ^= 1;
                        started = true;
                    } else {
                        fprintf(stderr, "WARNING: merged batch of %d families could not be enqueued (%s): running them one by one\n", G2,
                                gdca_last_error(cs[0]));
                    }
                }
                if (!started) {
                    // (no contexts for the sets, or the batch was refused: each family through the ordinary slots, synchronously)
                    for (Family &fm : grp) {
                        Slot &sl = slots[0];
                        if (sl.busy) finish(sl);
                        fill_slot(sl, fm, now());
                        gdca_params p{o.pseudocount, o.theta, o.score, 1};
                        if (gdca_run_ranked_async(sl.ctx, fm.Z(), fm.N, fm.M, fm.q, &p, o.min_separation) != GDCA_OK) {
                            fprintf(stderr, "ERROR: %s: gdca_run failed: %s\n", fm.name.c_str(), gdca_last_error(sl.ctx));
                            ++failures;
                            continue;
                        }
                        sl.busy = true;
                        finish(sl);
                    }
                }
                --k;  // (the big families' slots were not touched)
                continue;
            }
            Slot &sl = slots[k % (size_t)K];
            if (sl.busy) finish(sl);  // (round robin: the slot needed next holds the oldest run)
            const double t = now();
            fill_slot(sl, f, t);
            gdca_params p{o.pseudocount, o.theta, o.score, 1};
            const gdca_status rc = gdca_run_ranked_async(sl.ctx, f.Z(), f.N, f.M, f.q, &p, o.min_separation);
            busy_s += now() - t;
            if (rc != GDCA_OK) {
                fprintf(stderr, "ERROR: %s: gdca_run failed: %s\n", f.name.c_str(), gdca_last_error(sl.ctx));
                ++failures;
                --k;
                continue;
            }
            sl.busy = true;  // (the family's host matrix is no longer needed: it goes back to the reader's pool here)
        }
        drain(0);
        {
            std::lock_guard<std::mutex> lk(omu);
            busy[(size_t)g] += busy_s;
            count[(size_t)g] += done;
        }
        for (Set &st : sets)
            for (int m = (int)st.mem.size() - 1; m >= 0; --m) gdca_ctx_destroy(st.mem[(size_t)m].ctx);  // (peers before their leader)
        for (int k = K - 1; k >= 0; --k) gdca_ctx_destroy(slots[(size_t)k].ctx);
    };
    std::vector<std::thread> threads, writers;
    live_workers = G;
    for (int p = 0; p < std::max(1, o.parsers); ++p) threads.emplace_back(parser);
    for (int g = 0; g < G; ++g) threads.emplace_back(worker, g);
    for (int w = 0; w < std::max(2, 2 * G); ++w) writers.emplace_back(writer);
    for (auto &t : threads) t.join();
    {
        std::lock_guard<std::mutex> lk(omu);
        workers_done = true;
    }
    cv_out.notify_all();
    for (auto &t : writers) t.join();
    const double wall = now() - t0;
    if (abort_parsers) fprintf(stderr, "ERROR: no GPU worker could start; %zu families not processed\n", jobs.size());
    fprintf(stderr, "batch: %zu families on %d GPU(s) in %.3f s = %.2f families/s (%d failed)\n", jobs.size(), G, wall,
            (double)jobs.size() / wall, failures.load());
    {
        // the families that completed after the first `skip` did (one pipeline's worth: the contexts in flight, or the first phase batch --
        // its members all complete at once, which as "first to last" would count fifteen families at no time at all)
        const size_t skip = (size_t)std::max(1, o.inflight) * (size_t)G;
        if (done_at.size() > skip + 1 && done_at.back() > done_at[skip - 1])
            fprintf(stderr, "  steady state (after the first %zu completed families; process start, HIP initialisation and the first parses excluded): %.2f families/s%s\n",
                    skip, (double)(done_at.size() - skip) / (done_at.back() - done_at[skip - 1]),
                    merged_batches.load() > 0 ? "  (small families completed in phase batches, several at a time)" : "");
    }
    for (int g = 0; g < G; ++g) fprintf(stderr, "  gpu %d: %d families, busy %.3f s\n", devs[(size_t)g], count[(size_t)g], busy[(size_t)g]);
    return failures.load() ? 1 : 0;
}

}  // namespace

int main(int argc, char **argv)
{
    Options o;
    for (int a = 1; a < argc; ++a) {
        const std::string s = argv[a];
        auto val = [&]() -> const char * {
            if (a + 1 >= argc) die("missing value after " + s);
            return argv[++a];
        };
        if (s == "--pseudocount") o.pseudocount = atof(val());
        else if (s == "--theta") {
            const std::string v = val();
            o.theta = (v == "auto" || v == ":auto") ? -1.0 : atof(v.c_str());
            if (!(v == "auto" || v == ":auto") && o.theta < 0) die("invalid theta value: " + v + " (must be either :auto, or a number between 0 and 1)");
        } else if (s == "--max_gap_fraction") o.max_gap_fraction = atof(val());
        else if (s == "--score") {
            const std::string v = val();
            if (v == "frob" || v == ":frob") o.score = GDCA_SCORE_FROB;
            else if (v == "DI" || v == ":DI") o.score = GDCA_SCORE_DI;
            else die("invalid score value: " + v + " (must be either :DI or :frob)");
        } else if (s == "--min_separation") o.min_separation = atoi(val());
        else if (s == "--remove_dups") o.remove_dups = true;
        else if (s == "--batch") o.batch_dir = val();
        else if (s == "--out") o.out_dir = val();
        else if (s == "--gpus") o.gpus = atoi(val());
        else if (s == "--parsers") o.parsers = atoi(val());
        else if (s == "--inflight") o.inflight = atoi(val());
        else if (s == "--merge") {
            o.merge = std::min(32, std::max(1, atoi(val())));
            o.merge_given = true;
        }
        else if (s == "--merge-blocks") o.merge_blocks = std::min(1 << 20, std::max(1, atoi(val())));  // (0 or less: 1; "all": any number beyond the largest family)
        else if (s == "--parse-only") o.parse_only = true;
        else if (s == "--passes") o.passes = atoi(val());
        else if (s == "--synth") {
            if (a + 4 >= argc) die("usage: --synth N M SEED out.fasta[.gz]");
            const int N = atoi(argv[a + 1]), M = atoi(argv[a + 2]);
            const uint64_t seed = strtoull(argv[a + 3], nullptr, 0);
            std::vector<int8_t> Z((size_t)std::max(N, 1) * std::max(M, 1));
            if (gdca_synth_family(N, M, 21, seed, Z.data()) != GDCA_OK) die("invalid --synth sizes");
            if (gdca_write_fasta(argv[a + 4], Z.data(), N, M) != GDCA_OK) die(std::string("cannot write ") + argv[a + 4]);
            return 0;
        } else if (s == "-h" || s == "--help") {
            printf("usage: gdca_cli [--pseudocount X] [--theta auto|X] [--max_gap_fraction X] [--score frob|DI]\n"
                   "                [--min_separation K] [--remove_dups] alignment.fasta[.gz] [ranking.txt]\n"
                   "       gdca_cli [options] --batch DIR --out OUTDIR [--gpus G] [--parsers P] [--inflight K] [--merge K] [--merge-blocks B]\n"
                   "       gdca_cli [options] --batch DIR --parse-only [--parsers P] [--passes R]\n"
                   "       gdca_cli --synth N M SEED out.fasta[.gz]\n");
            return 0;
        } else if (!s.empty() && s[0] == '-' && s.size() > 1) die("unknown option " + s);
        else o.positional.push_back(s);
    }
    check_arguments(o);
    // parser threads: an eighth of a big host (32 at most), three quarters of a small one or of a small CPU quota
    if (o.parsers <= 0) {
        const int cpus = std::max(1, gdca_host_cpus());
        o.parsers = cpus >= 64 ? std::min(32, cpus / 8) : std::max(2, cpus * 3 / 4);
    }
    if (!o.batch_dir.empty()) {
        if (o.out_dir.empty() && !o.parse_only) die("--batch needs --out OUTDIR");
        return run_batch(o);
    }
    if (o.positional.empty()) die("no alignment given (see --help)");
    Family f;
    f.path = o.positional[0];
    struct stat sb;
    if (stat(f.path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) die("cannot open file " + f.path);
    if (!load_family(o, f)) die(f.error);
    gdca_ctx *ctx = nullptr;
    if (gdca_ctx_create(0, &ctx) != GDCA_OK) die("no usable HIP device (there is no CPU fallback)");
    gdca_stats st{};
    std::string err;
    const bool ok = process(ctx, o, f, o.positional.size() > 1 ? o.positional[1] : std::string(), &st, &err);
    if (ok)
        fprintf(stderr, "theta = %.16g threshold = %d\nM = %d N = %d Meff = %.16g\ndevice %.2f ms (inverse %.2f ms)\n", st.theta,
                st.thresh, f.M, f.N, st.Meff, st.ms_total, st.ms_inverse);
    else
        fprintf(stderr, "ERROR: %s\n", err.c_str());
    gdca_ctx_destroy(ctx);
    return ok ? 0 : 1;
}
