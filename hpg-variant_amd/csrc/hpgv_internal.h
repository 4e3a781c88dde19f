// hpgv_internal.h -- the context behind include/hpgv.h and the helpers shared by the translation units of
// libhpgv.so (hpgv_capi.hip: association / TDT / stats / text / batch entry points; hpgv_epi_capi.hip: the
// epistasis entry points, whose many kernel instantiations compile on their own).
#pragma once
#include "../../include/hpgv.h"
#include "hpgv_kernels.h"
#include "hpgv_tdt_stats_kernels.h"
#include "hpgv_epi_kernels.h"

#include <hip/hip_runtime.h>
#include <cctype>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace { thread_local std::string g_create_error; }

struct Layout {
    bool set = false;
    int n_samples = 0;
    size_t pitch = 0;
    int chunks = 0;
    std::vector<int32_t> col_of_pos;   // size pitch; -1 = pad
    int32_t *d_col_of_pos = nullptr;
    size_t d_cap = 0;                  // bytes behind d_col_of_pos
};

// per-call scratch of the synchronous host entry points
struct Slot {
    bool busy = false;
    hipStream_t stream = nullptr;
    void *buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void *h_res = nullptr;             // page-locked result block of the fused per-batch kernel (the kernel stores into it)
    void *d_res = nullptr;             // its device-side address
    size_t res_cap = 0;
    void *cnt_buf = nullptr;           // k_stats_all2's per-row counters between its two kernels
    size_t cnt_cap = 0;
};

// epistasis / MDR state: the vcf2epi dataset on the device, its bit planes for the current folds
struct EpiState {
    bool have_data = false, have_folds = false;
    int V = 0, nA = 0, nU = 0, num_folds = 0, W = 0, V_alloc = 0, n_chunks = 0;
    uint8_t *d_data = nullptr;
    uint32_t *d_planes = nullptr;
    uint32_t rev_off = 0;             // words from the planes to their copy with the low seven bits of every byte reversed (0: none; k_epi_pairs_mfma's column side)
    uint32_t *d_marg = nullptr;       // per SNP and (fold, class) group: samples with genotype 0 / 1 (16 bits each)
    bool complete = false;            // the dataset holds no call other than 0 / 1 / 2
    hpgv::EpiChunk *d_chunks = nullptr;
    uint32_t *d_chunk_cls = nullptr;  // per staging block: bit k = its step k holds controls
    hpgv::EpiFold *d_folds = nullptr;
    uint32_t *d_group_w0 = nullptr;
    std::vector<int32_t> group_size;
    hpgv::EpiCand *d_cand = nullptr;
    hpgv::EpiCand3 *d_cand3 = nullptr;    // the triple ranking's candidate lists, kept between calls
    size_t cand3_cap = 0;
    unsigned *d_cand_count = nullptr;
    unsigned cand_cap = 0;
    double *d_thr = nullptr;
    unsigned *d_tile_base = nullptr;
    size_t tile_base_cap = 0;
};

struct hpgv_ctx {
    // ---- a GROUP context (hpgv_create_multi) has members and nothing else: one ordinary context per device.  Cohort
    // calls go to every member, the synchronous per-batch calls to the member with the fewest calls in flight, the
    // device-resident calls to member 0; a member's failure text is copied to its group.
    std::vector<hpgv_ctx *> members;
    hpgv_ctx *parent = nullptr;
    struct GroupState *grp = nullptr;   // streams, scratch and the RCCL communicator of the group-wide resident scans (hpgv_group_capi.hip)
    long group_self_exchange = 0;       // test switch: member 0 also hands its results over through the communicator (send / recv to itself)
    std::atomic<unsigned> deal_next{0};
    std::atomic<int> in_flight{0};
    int device = 0;
    mutable std::string err;
    std::mutex mu;
    // options
    long row_align = 16;
    long row_pad = 0;          // extra bytes (multiple of 16) appended to every row; pitch exploration knob
    long vpw = 2;
    long nontemporal = 1;
    long profile = 0;
    std::mutex alias_mu;
    std::vector<std::pair<const char *, const char *>> text_alias;   // host text buffer -> the same text already on the device
    struct TextTiles { const char *d_text, *d_base; const void *d_tiles; uint64_t n_tiles; };
    std::vector<TextTiles> text_tiles;                               // device windows whose text comes with the decoder's tile records (hpgv_text_alias_tiles)
    long scan_unroll = 4;
    long persistent = 0;       // 0: one wave per vpw consecutive rows; 1: persistent strided grid
    long blocks_per_cu = 8;
    long pipeline = 1;         // 1: software-pipelined scan (loads of the next tile before counting this one)
    long pipe_waves = 4;       // register budget of the pipelined scan, as waves per SIMD (4, 6 or 8)
    long fisher_cut_exp = 22;  // Fisher tails stop after a round whose terms are all below 10^-this of the tail's largest term
    long epi_complete = 1;     // epistasis pair scan on a dataset without missing calls: count four cells, derive the other five
    long epi_pairs_mfma = 1;   // epistasis pair ranking, <= 10 folds, data with missing calls: cell counts on the matrix cores (k_epi_pairs_mfma); 0 = k_epi_pairs
    long epi_triples_mfma = 1; // epistasis triple ranking, <= 10 folds: cell counts on the matrix cores (k_epi_triples_mfma); 0 = the vector-ALU scans below
    long epi_triples_1pass = 1; // epistasis triple ranking with at most 10 folds: 1 = the 27 cells nine at a time (three walks, three waves per SIMD); 0 = the two-pass kernel; 2 (ablation build) = one pass with all counts in one lane
    long scan_lds = 0;         // bytes of (unused) LDS per workgroup of the stats / tdt scans: caps the waves in flight per CU
    long fisher_width = 16;    // lanes per variant in the Fisher p-pass (64, 32, 16 or 8): 64 / width variants per wave
    long inflate_wave = 1;     // bgzip decoder: 2 = one wave per block (hpgv_inflate2_kernels.h), 0 = one lane per block, 1 = by the number of blocks
    long tokenizer_tiles = 1;  // VCF text tokenizer: 1 = tile-parallel, two sweeps (count, scan, parse: the fastest); 2 = ONE sweep, the segments' states by look-back (k_tok_parse3: reads the text once, a third slower); 0 = count / mark / parse per line
    long batch_copy = 0;       // per-batch host entry points: 1 = copy page-locked rows to the device first (copy engine) instead of reading them in place
    long batch_fused = 1;      // per-batch host entry points: one fused kernel per call (0: copy + layout + scan + statistics kernels)
    long batch_lds_max = 65536;   // largest raw-row window the fused kernel stages in LDS (raised at hpgv_create when the device allows)
    // switches read ONCE from the environment at hpgv_create (include/hpgv.h "Environment"); no entry point reads the environment
    long stats_all2 = 1;       // HPGV_STATS_ALL2=0: every stats batch through the row-staging kernel (what shapes k_stats_all2 does not take use anyway)
    long assoc_rows = 1;       // HPGV_ASSOC_ROWS=0: text batches counted one workgroup per row (what wider cohorts fall back to)
    long pinned_noncoherent = 0;   // HPGV_PINNED_NONCOHERENT=1: page-locked buffers allocated non-coherent
    long vmm_trace = 0;        // HPGV_VMM_TRACE=1: hpgv_dev_commit narrates its mappings on stderr
    long decode_tiles = 1;     // HPGV_DECODE_TILES=0: windows of device-decoded text are tokenized with the counting sweep even when the decoder left its tile records
#ifdef HPGV_ABLATION
    long stats_rows = 0, stats_bs = 0, stats_debug = 0;      // HPGV_STATS_ROWS / _BS / _DEBUG: band length, workgroup size, chosen form of k_stats_all2
    long fisher_lds = 0;       // HPGV_FISHER_LDS: unused LDS bytes per workgroup of the Fisher pass
    long inflate_lds_pad = 0, inflate_wave_wgs = 0, inflate_lane_wgs = 0;      // HPGV_INFLATE_*: the decoder's occupancy experiments
#endif
    int n_cus = 256;
    // assoc
    Layout assoc;
    int nA = 0, nU = 0, chunksA = 0;
    uint8_t *d_cond = nullptr;            // the condition of every column as given (padded with 2 to whole 16-byte chunks): k_assoc_rows' masks
    size_t cond_cap = 0;
    // tdt
    Layout tdt;
    hpgv::TdtPlan tdt_plan;
    // stats
    Layout stats;
    Layout sgroups;                       // [group 0 | pad16 | group 1 | ...]
    std::vector<uint32_t> sg_off;         // byte offset of every group's segment in the row
    std::vector<int> sg_size;             // samples per group
    int32_t *d_sg_chunks = nullptr;       // device: first 16-byte chunk and chunk count of every group ([2 * n_groups])
    size_t sg_chunks_cap = 0;
    uint8_t *d_group_of_col = nullptr;    // device: group id of every column (0xFF: in no group), padded to whole 16-byte chunks -- k_stats_all2's masks
    size_t group_of_col_cap = 0;
    bool all_grouped = false;             // every column is in a group: the last group's counters are "all minus the others"
    // mendelian errors
    Layout mendel;
    int mendel_trios = 0, mendel_pchunks = 0;
    hpgv::MendelLuts mendel_luts{};
    uint8_t *d_mendel_male = nullptr;
    // fisher
    double *d_lf = nullptr;
    double *d_lf_base = nullptr;       // the allocation: d_lf - 2 (padding for the two-entries-per-load reads of the Fisher pass)
    size_t n_lf = 0;
    size_t cap_lf = 0;                 // doubles behind d_lf (kept across tables: hipFree waits for the whole device)
    // synth scratch
    uint32_t *d_thr = nullptr;
    size_t thr_cap = 0;
    // profiling
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool have_scan_ev = false, have_stats_ev = false;
    std::vector<Slot *> slots;
    uint32_t *d_sink = nullptr;
    uint32_t *d_crc_tab = nullptr;     // tables of the BGZF CRC-32 check (hpgv_crc_kernels.h), built at first use
    // tokenizer scratch (newline counts per 4 KiB tile, line offsets), one set per stream that has
    // tokenized: calls on one stream are ordered by the stream, calls on different streams run
    // concurrently on the device and must not share it.  The table is guarded by tok_mu.
    struct TokScratch {
        hipStream_t stream = nullptr;
        int *d_blocks = nullptr; size_t blocks_cap = 0;
        unsigned long long *d_line_off = nullptr; size_t line_cap = 0;
        void *d_extra = nullptr; size_t extra_cap = 0;
    };
    std::mutex tok_mu;
    std::vector<TokScratch *> tok_scratch;
    // address ranges whose backing grows (hpgv_dev_reserve / hpgv_dev_commit), under mu
    struct GrowRange { char *base = nullptr; size_t reserved = 0, committed = 0; std::vector<hipMemGenericAllocationHandle_t> pieces; std::vector<size_t> sizes; };
    std::vector<GrowRange> grow;
    // record filters of the text entry points (hpgv_set_text_filters); negative = off
    double filt_min_maf = -1.0, filt_max_missing = -1.0;
    long filt_max_mendel = -1;
    // epistasis (calls are serialised by epi_mu)
    std::mutex epi_mu;
    EpiState epi;
};

namespace {

[[maybe_unused]] int fail(const hpgv_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) {
        ctx->err = buf;
        if (ctx->parent) {
            std::lock_guard<std::mutex> lk(ctx->parent->mu);
            ctx->parent->err = buf;
        }
    } else g_create_error = buf;
    return code;
}

// the C ABI never lets a C++ exception out: a failed host allocation inside an entry point (std::vector, std::string)
// becomes HPGV_ERR_NOMEM.  The slot lease and the device guard are released by their destructors during unwinding.
#define HPGV_ABI_TRY try {
#define HPGV_ABI_CATCH(ctx)                                                                         \
    } catch (const std::bad_alloc &) { return fail(ctx, HPGV_ERR_NOMEM, "out of host memory"); }     \
      catch (...) { return fail(ctx, HPGV_ERR_INVALID, "unexpected C++ exception inside the engine"); }

// group dispatch
inline bool is_group(const hpgv_ctx *c) { return c && !c->members.empty(); }
inline hpgv_ctx *first_member(hpgv_ctx *c) { return is_group(c) ? c->members[0] : c; }
inline const hpgv_ctx *first_member(const hpgv_ctx *c) { return is_group(c) ? c->members[0] : c; }
// the member a synchronous per-batch call runs on: fewest calls in flight, round-robin among equals
struct Dealt {
    hpgv_ctx *m;
    explicit Dealt(hpgv_ctx *g) {
        const unsigned n = (unsigned)g->members.size(), start = g->deal_next.fetch_add(1u, std::memory_order_relaxed) % n;
        m = g->members[start];
        int best = m->in_flight.load(std::memory_order_relaxed);
        for (unsigned k = 1; k < n && best > 0; ++k) {
            hpgv_ctx *c = g->members[(start + k) % n];
            const int f = c->in_flight.load(std::memory_order_relaxed);
            if (f < best) { best = f; m = c; }
        }
        m->in_flight.fetch_add(1, std::memory_order_relaxed);
    }
    ~Dealt() { m->in_flight.fetch_sub(1, std::memory_order_relaxed); }
};
#define GROUP_ALL(ctx, CALL)                                                                \
    if (is_group(ctx)) {                                                                    \
        for (hpgv_ctx *m_ : (ctx)->members) { const int rc_ = CALL; if (rc_) return rc_; }  \
        return HPGV_OK;                                                                     \
    }
#define GROUP_DEAL(ctx, CALL)                                                               \
    if (is_group(ctx)) { Dealt d_(ctx); hpgv_ctx *m_ = d_.m; return CALL; }

#define HIPCHK(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(ctx, HPGV_ERR_HIP, "%s failed: %s (%s:%d)", #call,                  \
                        hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

// makes ctx->device current for the scope of one API call
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            changed = (hipSetDevice(dev) == hipSuccess);
        }
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};

[[maybe_unused]] size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

[[maybe_unused]] int upload_layout(hpgv_ctx *ctx, Layout &L) {
    // the table is kept when it is large enough: hipFree waits for every stream of the device, and a file run sets its
    // cohort while the decoder of the bgzip text is busy on streams of its own
    const size_t need = L.col_of_pos.size() * sizeof(int32_t);
    if (L.d_cap < need) {
        if (L.d_col_of_pos) { (void)hipFree(L.d_col_of_pos); L.d_col_of_pos = nullptr; L.d_cap = 0; }
        HIPCHK(ctx, hipMalloc(&L.d_col_of_pos, need));
        L.d_cap = need;
    }
    HIPCHK(ctx, hipMemcpy(L.d_col_of_pos, L.col_of_pos.data(), L.col_of_pos.size() * sizeof(int32_t),
                          hipMemcpyHostToDevice));
    L.chunks = (int)(L.pitch / 16);
    L.set = true;
    return HPGV_OK;
}

// packed per-lane 16-bit partial sums bound the row length (hpgv_kernels.h)
constexpr int kScanUnroll = 8;       // unroll of the tdt/stats scans
constexpr int kMaxUnroll = 16;       // largest assoc unroll option
[[maybe_unused]] bool pitch_supported(size_t pitch) { return pitch / 16 / 64 + kMaxUnroll + 1 <= 2047; }

[[maybe_unused]] int ensure(hpgv_ctx *ctx, Slot *s, int idx, size_t bytes) {
    if (s->cap[idx] >= bytes) return HPGV_OK;
    if (s->buf[idx]) { (void)hipFree(s->buf[idx]); s->buf[idx] = nullptr; s->cap[idx] = 0; }
    size_t want = round_up(bytes + bytes / 4, 256);
    HIPCHK(ctx, hipMalloc(&s->buf[idx], want));
    s->cap[idx] = want;
    return HPGV_OK;
}

[[maybe_unused]] int acquire_slot(hpgv_ctx *ctx, Slot **out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (Slot *s : ctx->slots)
        if (!s->busy) { s->busy = true; *out = s; return HPGV_OK; }
    Slot *s = new (std::nothrow) Slot();
    if (!s) return fail(ctx, HPGV_ERR_NOMEM, "out of host memory");
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete s; return fail(ctx, HPGV_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    s->busy = true;
    ctx->slots.push_back(s);
    *out = s;
    return HPGV_OK;
}
[[maybe_unused]] void release_slot(hpgv_ctx *ctx, Slot *s) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    s->busy = false;
}
struct SlotLease {
    hpgv_ctx *ctx; Slot *s = nullptr;
    explicit SlotLease(hpgv_ctx *c) : ctx(c) {}
    // an early return on a failure may leave copies into the caller's (or this call's stack) memory queued on the slot's
    // stream: they are waited for before the slot -- and the caller's buffers -- are handed back
    ~SlotLease() {
        if (!s) return;
        if (s->stream && hipStreamQuery(s->stream) == hipErrorNotReady) (void)hipStreamSynchronize(s->stream);
        (void)hipGetLastError();
        release_slot(ctx, s);
    }
};

template <typename F>
[[maybe_unused]] int launch_profiled(hpgv_ctx *ctx, hipStream_t st, int which /*0 scan,1 stats*/, F &&launch) {
    if (ctx->profile) HIPCHK(ctx, hipEventRecord(ctx->ev[2 * which], st));
    launch();
    HIPCHK(ctx, hipGetLastError());
    if (ctx->profile) {
        HIPCHK(ctx, hipEventRecord(ctx->ev[2 * which + 1], st));
        (which == 0 ? ctx->have_scan_ev : ctx->have_stats_ev) = true;
    }
    return HPGV_OK;
}

}  // namespace


// defined in hpgv_statsall_capi.hip: k_stats_all2 on a batch (0 = launched, 1 = not a batch it takes: run k_stats_all)
namespace hpgv { struct StatsAllArgs; }
int hpgv_launch_stats_all2(hpgv_ctx *ctx, hpgv::StatsAllArgs &A, void **cnt_buf, size_t *cnt_cap, hipStream_t st);
// k_assoc_rows: the allele counts from the tokenizer's raw rows (0 = launched, 1 = not a batch it takes: run k_batch)
int hpgv_launch_assoc_rows(hpgv_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int n_variants, const uint8_t *d_is_x, int32_t *d_counts, hipStream_t st);
// defined in hpgv_epi_capi.hip
void hpgv_epi_release(EpiState &E);
// defined in hpgv_epi_generic_capi.hip: in-fold counts of listed combinations of order 2 .. 5, device to device (the caller holds epi_mu)
int hpgv_epi_generic_counts(hpgv_ctx *ctx, int order, const int32_t *d_combs, int n_combs, int32_t *d_out);
// defined in hpgv_group_capi.hip: streams, scratch and communicator of a group context (before its members go)
void hpgv_group_release(hpgv_ctx *group);
