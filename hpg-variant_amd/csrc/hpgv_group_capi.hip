// hpgv_group_capi.hip -- the variant-sharded resident scan of a group context (include/hpgv.h "hpgv_group_*").
//
// The reference's runner fans batches of variants out to its workers and collects one record per variant
// (assoc_runner.c:106-207, tdt_runner.c:150-200, stats_runner.c:176-215); variants carry no state from one to the next
// (assoc.c:38-82, tdt.c:41-271).  With the cohort resident in the HBM of G devices this becomes: member g scans the
// contiguous shard [g*V/G, (g+1)*V/G) on a stream of its own -- no traffic between devices while scanning -- and the ONE
// exchange is the gather of the per-variant result pieces onto member 0: grouped ncclSend / ncclRecv on a
// communicator the group owns (ncclCommInitAll over the members' devices: one process, RCCL over xGMI, every peer on its
// own link into member 0).  Per-sample counters (get_sample_stats) are sums over variants: ncclReduce onto member 0.
//
// librccl is loaded with dlopen when the communicator is first asked for: libhpgv.so itself has no RCCL dependency, a
// single-device user never loads it, and inside a process that already holds an RCCL (torch's) the same copy is used.
//
// Streams: every member has a scan stream and a transfer stream.  A call queues, per member: [wait until the transfer
// that last read this generation's scratch is done] scan + statistics kernels -> event -> (transfer stream) send of the
// pieces.  Two generations of scratch per member let the transfers of call k run under the scans of call k + 1.
#include "hpgv_internal.h"

#include <dlfcn.h>
#include <thread>
// types and enums only: every function is reached through dlsym.  A build box without the RCCL headers still builds the
// library: the few declarations the group scan uses are restated below (values as in rccl.h; RCCL keeps them ABI-stable).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclInt32 = 2 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count);
const char *ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream);
}
#endif

struct GroupMember {
    hipStream_t scan = nullptr, xfer = nullptr;
    hipEvent_t scan_done[2] = {nullptr, nullptr}, xfer_done[2] = {nullptr, nullptr};
    bool xfer_pending[2] = {false, false};
    void *scratch[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
    int32_t *d_miss[2] = {nullptr, nullptr};   // per-sample counters of this member's shard, per generation like the scratch
    size_t miss_cap[2] = {0, 0};
    int rank = 0;                   // RCCL rank of the member's device
    bool local = false;             // shares member 0's device: results handed over by a device-local copy
};

struct GroupState {
    std::mutex mu;                  // one group call is queued at a time
    void *dl = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    std::vector<ncclComm_t> comms;  // one per distinct device, rank r = r-th distinct device in member order
    std::vector<int> devs;
    std::vector<GroupMember> m;
    unsigned gen = 0;
    bool ready = false;
};

namespace {

// several contexts of member 0's device add their counters at the same time, and member 0's own scan may still be counting
__global__ void k_add_i32(int32_t *__restrict__ dst, const int32_t *__restrict__ src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && src[i]) atomicAdd(&dst[i], src[i]);
}

#define NCCLCHK(g, S, call)                                                                       \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess)                                                                    \
            return fail(g, HPGV_ERR_HIP, "%s failed: %s (%s:%d)", #call, (S)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// dlopen of librccl under its usual names ($HPGV_RCCL_LIB first); `tried` collects why each candidate failed
void *open_rccl(std::string &tried) {
    // The communicator must sit on the SAME HIP / HSA runtime this library is bound to.  A process may hold a second ROCm stack
    // (importing torch after this library loads torch's bundled copies beside /opt/rocm's): a bare dlopen("librccl.so.1") then
    // hands back whichever librccl is already loaded, and one bound to the other stack finds its HSA uninitialised
    // ("no ROCm-capable device").  So the librccl NEXT TO the HIP runtime in use is tried first, by full path.
    std::string beside1, beside2;
    {
        Dl_info info;
        if (dladdr((const void *)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) { dir.resize(slash + 1); beside1 = dir + "librccl.so.1"; beside2 = dir + "librccl.so"; }
        }
    }
    const char *env = getenv("HPGV_RCCL_LIB");
    const char *names[] = {env, beside1.c_str(), beside2.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        void *dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (dl) return dl;
        const char *e = dlerror();                      // ONE call: dlerror() clears the message it returns
        tried += std::string(tried.empty() ? "" : "; ") + std::string(e ? e : n);
    }
    return nullptr;
}

int load_rccl(hpgv_ctx *g, GroupState *S) {
    if (S->dl) return HPGV_OK;
    std::string tried;
    S->dl = open_rccl(tried);
    if (!S->dl)
        return fail(g, HPGV_ERR_UNSUPPORTED, "the group-wide scan gathers its results over RCCL and librccl could not be loaded (%s); "
                                             "set HPGV_RCCL_LIB to its path", tried.c_str());
#define SYM(field, name)                                                                          \
    S->field = (decltype(S->field))dlsym(S->dl, name);                                            \
    if (!S->field) { dlclose(S->dl); S->dl = nullptr; return fail(g, HPGV_ERR_UNSUPPORTED, "librccl lacks %s", name); }
    SYM(CommInitAll, "ncclCommInitAll") SYM(CommDestroy, "ncclCommDestroy") SYM(CommCount, "ncclCommCount")
    SYM(GetErrorString, "ncclGetErrorString") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
    SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(Reduce, "ncclReduce")
#undef SYM
    return HPGV_OK;
}

void shard_of(int64_t V, int G, int g, int64_t *lo, int64_t *hi) {
    *lo = (int64_t)((__int128)V * g / G);
    *hi = (int64_t)((__int128)V * (g + 1) / G);
}

int ensure_scratch(hpgv_ctx *mc, GroupMember &M, int gen, size_t bytes) {
    if (M.cap[gen] >= bytes) return HPGV_OK;
    if (M.scratch[gen]) { (void)hipFree(M.scratch[gen]); M.scratch[gen] = nullptr; M.cap[gen] = 0; }
    const size_t want = round_up(bytes + bytes / 16, 256);
    HIPCHK(mc, hipMalloc(&M.scratch[gen], want));
    M.cap[gen] = want;
    return HPGV_OK;
}

// one result piece: `elem` bytes per variant, gathered into dst (member 0's device, variant v at dst + v * elem)
struct Piece { size_t elem; void *dst; };

struct Plan {
    hpgv_ctx *g;
    GroupState *S;
    int G, gen;
    int64_t V;
    std::vector<int64_t> lo, n;
    std::vector<bool> via_scratch;      // member's results go to its scratch first and are handed over
    std::vector<char *> base;           // where member k's piece 0 starts (scratch, or the destination itself for member 0)
};

// validates, cuts the shards, makes the scratch of this generation ready and the scan stream wait for the transfer that
// last read it
int plan_call(hpgv_ctx *g, int64_t V, size_t bytes_per_variant, Plan &P) {
    if (!is_group(g)) return fail(g, HPGV_ERR_INVALID, "hpgv_group_* needs a group context (hpgv_create_multi)");
    if (V < 0) return fail(g, HPGV_ERR_INVALID, "n_variants < 0");
    GroupState *S = g->grp;
    P.g = g; P.S = S; P.G = (int)g->members.size(); P.V = V;
    P.gen = (int)(S->gen++ & 1u);
    P.lo.resize(P.G); P.n.resize(P.G); P.via_scratch.resize(P.G); P.base.assign(P.G, nullptr);
    for (int k = 0; k < P.G; ++k) {
        int64_t lo, hi;
        shard_of(V, P.G, k, &lo, &hi);
        if (hi - lo > 0x7fffffff) return fail(g, HPGV_ERR_UNSUPPORTED, "a member's shard has more than 2^31 - 1 variants");
        P.lo[k] = lo; P.n[k] = hi - lo;
        P.via_scratch[k] = k > 0 || g->group_self_exchange;
        hpgv_ctx *mc = g->members[k];
        GroupMember &M = S->m[k];
        DeviceGuard dg(mc->device);
        // the transfer of two calls ago read this generation's scratch (member 0: wrote the caller's arrays of that call,
        // which a caller alternating between two result sets hands in again now)
        if (M.xfer_pending[P.gen]) {
            HIPCHK(mc, hipStreamWaitEvent(M.scan, M.xfer_done[P.gen], 0));
            M.xfer_pending[P.gen] = false;
        }
        if (!P.via_scratch[k]) continue;
        const int e = ensure_scratch(mc, M, P.gen, (size_t)P.n[k] * bytes_per_variant + 256);
        if (e) return e;
        P.base[k] = (char *)M.scratch[P.gen];
    }
    return HPGV_OK;
}

// a call that fails after plan_call has kernels queued on members' scan streams and no event recorded for them: the
// streams are drained, so that the scratch of this generation and the caller's arrays are quiet when the error returns
int drain(const Plan &P, int rc) {
    for (int k = 0; k < P.G; ++k) {
        GroupMember &M = P.S->m[(size_t)k];
        DeviceGuard dg(P.g->members[(size_t)k]->device);
        if (M.scan) (void)hipStreamSynchronize(M.scan);
        if (M.xfer) (void)hipStreamSynchronize(M.xfer);
        M.xfer_pending[0] = M.xfer_pending[1] = false;
    }
    (void)hipGetLastError();
    return rc;
}

// address of member k's piece i (`before` = bytes per variant of the pieces in front of it)
inline char *piece_src(const Plan &P, int k, const Piece &pc, size_t before) {
    if (!P.via_scratch[k]) return (char *)pc.dst + (size_t)P.lo[k] * pc.elem;
    return P.base[k] + (size_t)P.n[k] * before;
}

// after the members' kernels are queued: the hand-over of the pieces onto member 0
int exchange(Plan &P, const std::vector<Piece> &pieces) {
    hpgv_ctx *g = P.g;
    GroupState *S = P.S;
    GroupMember &M0 = S->m[0];
    for (int k = 0; k < P.G; ++k) {
        hpgv_ctx *mc = g->members[k];
        GroupMember &M = S->m[k];
        DeviceGuard dg(mc->device);
        HIPCHK(mc, hipEventRecord(M.scan_done[P.gen], M.scan));
        if (P.via_scratch[k]) HIPCHK(mc, hipStreamWaitEvent(M.xfer, M.scan_done[P.gen], 0));
    }
    // how member k hands its pieces over: a context on another device sends them through the communicator; a second context
    // on member 0's device copies them (RCCL refuses one device twice: the one-GPU test rig); member 0 itself scans into the
    // destination -- or, with the test switch group_self_exchange, sends to itself through the communicator
    auto by_rccl = [&](int k) { return P.via_scratch[k] && (!S->m[(size_t)k].local || k == 0); };
    bool any_rccl = false;
    for (int k = 0; k < P.G; ++k) any_rccl = any_rccl || (by_rccl(k) && P.n[k] > 0);
    if (any_rccl) {
        NCCLCHK(g, S, S->GroupStart());
        for (int k = 0; k < P.G; ++k) {
            GroupMember &M = S->m[(size_t)k];
            if (!by_rccl(k) || P.n[k] == 0) continue;
            size_t before = 0;
            for (const Piece &pc : pieces) {
                if (pc.dst) {
                    const size_t bytes = (size_t)P.n[k] * pc.elem;
                    NCCLCHK(g, S, S->Send(piece_src(P, k, pc, before), bytes, ncclInt8, 0, S->comms[(size_t)M.rank], M.xfer));
                    NCCLCHK(g, S, S->Recv((char *)pc.dst + (size_t)P.lo[k] * pc.elem, bytes, ncclInt8, M.rank, S->comms[0], M0.xfer));
                }
                before += pc.elem;
            }
        }
        NCCLCHK(g, S, S->GroupEnd());
    }
    for (int k = 0; k < P.G; ++k) {
        GroupMember &M = S->m[(size_t)k];
        hpgv_ctx *mc = g->members[(size_t)k];
        DeviceGuard dg(mc->device);
        if (P.via_scratch[k] && !by_rccl(k) && P.n[k] > 0) {
            size_t before = 0;
            for (const Piece &pc : pieces) {
                if (pc.dst)
                    HIPCHK(mc, hipMemcpyAsync((char *)pc.dst + (size_t)P.lo[k] * pc.elem, piece_src(P, k, pc, before),
                                              (size_t)P.n[k] * pc.elem, hipMemcpyDeviceToDevice, M.xfer));
                before += pc.elem;
            }
        }
        if (P.via_scratch[k] || k == 0) {
            HIPCHK(mc, hipEventRecord(M.xfer_done[P.gen], M.xfer));
            M.xfer_pending[P.gen] = true;
        }
    }
    return HPGV_OK;
}

}  // namespace

void hpgv_group_release(hpgv_ctx *g) {
    if (!g || !g->grp) return;
    GroupState *S = g->grp;
    for (size_t k = 0; k < S->m.size() && k < g->members.size(); ++k) {
        GroupMember &M = S->m[k];
        DeviceGuard dg(g->members[k]->device);
        if (M.scan) (void)hipStreamSynchronize(M.scan);
        if (M.xfer) (void)hipStreamSynchronize(M.xfer);
    }
    if (S->CommDestroy)
        for (ncclComm_t c : S->comms) if (c) (void)S->CommDestroy(c);
    for (size_t k = 0; k < S->m.size() && k < g->members.size(); ++k) {
        GroupMember &M = S->m[k];
        DeviceGuard dg(g->members[k]->device);
        for (int i = 0; i < 2; ++i) {
            if (M.scratch[i]) (void)hipFree(M.scratch[i]);
            if (M.scan_done[i]) (void)hipEventDestroy(M.scan_done[i]);
            if (M.xfer_done[i]) (void)hipEventDestroy(M.xfer_done[i]);
        }
        for (int i = 0; i < 2; ++i) if (M.d_miss[i]) (void)hipFree(M.d_miss[i]);
        if (M.scan) (void)hipStreamDestroy(M.scan);
        if (M.xfer) (void)hipStreamDestroy(M.xfer);
    }
    // the library stays loaded: other groups (and the process's own RCCL users) may hold it
    delete S;
    g->grp = nullptr;
}

extern "C" {

int hpgv_group_comm_init(hpgv_ctx *g) {
    HPGV_ABI_TRY
    if (!is_group(g)) return fail(g, HPGV_ERR_INVALID, "hpgv_group_comm_init needs a group context (hpgv_create_multi)");
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->grp && g->grp->ready) return HPGV_OK;
    if (!g->grp) g->grp = new GroupState();
    GroupState *S = g->grp;
    const int G = (int)g->members.size();
    // ranks = the distinct devices in member order; a device may repeat only when it is member 0's
    S->devs.clear();
    S->m.assign((size_t)G, GroupMember());
    for (int k = 0; k < G; ++k) {
        const int dev = g->members[(size_t)k]->device;
        int r = -1;
        for (size_t i = 0; i < S->devs.size(); ++i) if (S->devs[i] == dev) r = (int)i;
        if (r > 0) { hpgv_group_release(g); return fail(g, HPGV_ERR_UNSUPPORTED, "device %d is listed twice and is not member 0's: only member 0's device may repeat (the one-GPU test rig)", dev); }
        if (r < 0) { r = (int)S->devs.size(); S->devs.push_back(dev); }
        S->m[(size_t)k].rank = r;
        S->m[(size_t)k].local = (r == 0);
    }
    int rc = load_rccl(g, S);
    if (rc) { hpgv_group_release(g); return rc; }
    S->comms.assign(S->devs.size(), nullptr);
    {
        ncclResult_t r = S->CommInitAll(S->comms.data(), (int)S->devs.size(), S->devs.data());
        if (r != ncclSuccess) {
            rc = fail(g, HPGV_ERR_HIP, "ncclCommInitAll over %d device(s) failed: %s", (int)S->devs.size(), S->GetErrorString(r));
            S->comms.clear();
            hpgv_group_release(g);
            return rc;
        }
    }
    for (int k = 0; k < G; ++k) {
        hpgv_ctx *mc = g->members[(size_t)k];
        GroupMember &M = S->m[(size_t)k];
        DeviceGuard dg(mc->device);
        hipError_t e = hipStreamCreateWithFlags(&M.scan, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&M.xfer, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; ++i) {
            e = hipEventCreateWithFlags(&M.scan_done[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&M.xfer_done[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            rc = fail(g, HPGV_ERR_HIP, "group streams on device %d: %s", mc->device, hipGetErrorString(e));
            hpgv_group_release(g);
            return rc;
        }
    }
    S->ready = true;
    return HPGV_OK;
    HPGV_ABI_CATCH(g)
}

int hpgv_group_rccl_probe(char *why, size_t why_cap) {
    try {
        std::string tried;
        void *dl = open_rccl(tried);
        if (why && why_cap) snprintf(why, why_cap, "%s", tried.c_str());
        if (!dl) return HPGV_ERR_UNSUPPORTED;
        const bool ok = dlsym(dl, "ncclCommInitAll") && dlsym(dl, "ncclSend") && dlsym(dl, "ncclRecv") && dlsym(dl, "ncclReduce");
        dlclose(dl);
        return ok ? HPGV_OK : HPGV_ERR_UNSUPPORTED;
    } catch (...) { return HPGV_ERR_NOMEM; }
}

int hpgv_group_comm_ranks(const hpgv_ctx *g) {
    if (!is_group(g)) return 0;
    // g->mu guards g->grp itself: hpgv_group_comm_init may delete the state on a failure path
    std::lock_guard<std::mutex> lk(const_cast<hpgv_ctx *>(g)->mu);
    if (!g->grp || !g->grp->ready || g->grp->comms.empty()) return 0;
    int n = 0;
    if (g->grp->CommCount(g->grp->comms[0], &n) != ncclSuccess) return 0;
    return n;
}

int hpgv_group_shard(const hpgv_ctx *g, int64_t n_variants, int member, int64_t *lo, int64_t *hi) {
    if (!g || !lo || !hi || n_variants < 0) return HPGV_ERR_INVALID;
    const int G = hpgv_group_size(g);
    if (member < 0 || member >= G) return HPGV_ERR_INVALID;
    shard_of(n_variants, G, member, lo, hi);
    return HPGV_OK;
}

int hpgv_group_sync(hpgv_ctx *g) {
    if (!is_group(g)) return fail(g, HPGV_ERR_INVALID, "hpgv_group_sync needs a group context");
    GroupState *S = nullptr;
    {   // g->mu guards g->grp itself (see hpgv_group_comm_ranks); a ready state lives until hpgv_destroy
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->grp && g->grp->ready) S = g->grp;
    }
    if (!S) return HPGV_OK;
    std::lock_guard<std::mutex> lk(S->mu);
    for (size_t k = 0; k < g->members.size(); ++k) {
        hpgv_ctx *mc = g->members[k];
        GroupMember &M = g->grp->m[k];
        DeviceGuard dg(mc->device);
        HIPCHK(g, hipStreamSynchronize(M.scan));
        HIPCHK(g, hipStreamSynchronize(M.xfer));
        M.xfer_pending[0] = M.xfer_pending[1] = false;
    }
    return HPGV_OK;
}

int hpgv_group_assoc(hpgv_ctx *g, int task, const uint8_t *const *d_gt, const uint8_t *const *d_is_x, int64_t V,
                     int32_t *d_counts, double *d_odds, double *d_chisq, double *d_p) {
    HPGV_ABI_TRY
    if (!is_group(g)) return fail(g, HPGV_ERR_INVALID, "hpgv_group_assoc needs a group context (hpgv_create_multi)");
    if (task != HPGV_TASK_CHISQ && task != HPGV_TASK_FISHER) return fail(g, HPGV_ERR_INVALID, "task must be HPGV_TASK_CHISQ or HPGV_TASK_FISHER");
    if (!d_gt || (V > 0 && (!d_counts || !d_odds || !d_p || (task == HPGV_TASK_CHISQ && !d_chisq))))
        return fail(g, HPGV_ERR_INVALID, "bad group assoc arguments");
    const bool chisq = task == HPGV_TASK_CHISQ;
    const std::vector<Piece> pieces = {{16, d_counts}, {8, d_odds}, {8, chisq ? d_chisq : nullptr}, {8, d_p}};
    int rc = hpgv_group_comm_init(g);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->grp->mu);
    Plan P;
    rc = plan_call(g, V, 40, P);
    if (rc) return rc;
    for (int k = 0; k < P.G; ++k) {
        if (P.n[k] == 0) continue;
        hpgv_ctx *mc = g->members[(size_t)k];
        GroupMember &M = P.S->m[(size_t)k];
        if (!d_gt[k]) return drain(P, fail(g, HPGV_ERR_INVALID, "member %d has %lld variants but no matrix", k, (long long)P.n[k]));
        const int n = (int)P.n[k];
        int32_t *c = (int32_t *)piece_src(P, k, pieces[0], 0);
        double *o = (double *)piece_src(P, k, pieces[1], 16);
        double *x = chisq ? (double *)piece_src(P, k, pieces[2], 24) : nullptr;
        double *p = (double *)piece_src(P, k, pieces[3], 32);
        rc = hpgv_assoc_scan_dev(mc, d_gt[k], n, d_is_x ? d_is_x[k] : nullptr, c, M.scan);
        if (!rc) rc = chisq ? hpgv_assoc_chisq_dev(mc, c, n, o, x, p, M.scan) : hpgv_assoc_fisher_dev(mc, c, n, o, p, M.scan);
        if (rc) return drain(P, rc);
    }
    rc = exchange(P, pieces);
    return rc ? drain(P, rc) : HPGV_OK;
    HPGV_ABI_CATCH(g)
}

int hpgv_group_tdt(hpgv_ctx *g, const uint8_t *const *d_gt, const uint8_t *const *d_is_x, int64_t V, int32_t *d_tu,
                   double *d_odds, double *d_chisq, double *d_p) {
    HPGV_ABI_TRY
    if (!is_group(g)) return fail(g, HPGV_ERR_INVALID, "hpgv_group_tdt needs a group context (hpgv_create_multi)");
    if (!d_gt || (V > 0 && (!d_tu || !d_odds || !d_chisq || !d_p))) return fail(g, HPGV_ERR_INVALID, "bad group tdt arguments");
    const std::vector<Piece> pieces = {{8, d_tu}, {8, d_odds}, {8, d_chisq}, {8, d_p}};
    int rc = hpgv_group_comm_init(g);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->grp->mu);
    Plan P;
    rc = plan_call(g, V, 32, P);
    if (rc) return rc;
    for (int k = 0; k < P.G; ++k) {
        if (P.n[k] == 0) continue;
        hpgv_ctx *mc = g->members[(size_t)k];
        GroupMember &M = P.S->m[(size_t)k];
        if (!d_gt[k]) return drain(P, fail(g, HPGV_ERR_INVALID, "member %d has %lld variants but no matrix", k, (long long)P.n[k]));
        const int n = (int)P.n[k];
        int32_t *tu = (int32_t *)piece_src(P, k, pieces[0], 0);
        rc = hpgv_tdt_scan_dev(mc, d_gt[k], n, d_is_x ? d_is_x[k] : nullptr, tu, M.scan);
        if (!rc) rc = hpgv_tdt_stats_dev(mc, tu, n, (double *)piece_src(P, k, pieces[1], 8), (double *)piece_src(P, k, pieces[2], 16),
                                         (double *)piece_src(P, k, pieces[3], 24), M.scan);
        if (rc) return drain(P, rc);
    }
    rc = exchange(P, pieces);
    return rc ? drain(P, rc) : HPGV_OK;
    HPGV_ABI_CATCH(g)
}

int hpgv_group_stats(hpgv_ctx *g, const uint8_t *const *d_gt, int64_t V, int32_t *d_counts8, double *d_hwe_chi2,
                     double *d_hwe_p, int32_t *d_sample_missing) {
    HPGV_ABI_TRY
    if (!is_group(g)) return fail(g, HPGV_ERR_INVALID, "hpgv_group_stats needs a group context (hpgv_create_multi)");
    if (!d_gt || (V > 0 && (!d_counts8 || !d_hwe_chi2 || !d_hwe_p))) return fail(g, HPGV_ERR_INVALID, "bad group stats arguments");
    const std::vector<Piece> pieces = {{32, d_counts8}, {8, d_hwe_chi2}, {8, d_hwe_p}};
    int rc = hpgv_group_comm_init(g);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->grp->mu);
    Plan P;
    rc = plan_call(g, V, 48, P);
    if (rc) return rc;
    GroupState *S = P.S;
    const int n_samples = g->members[0]->stats.n_samples;
    // everything queued from here on is one unit: any failure inside it drains the members' streams before it returns
    const auto queued = [&]() -> int {
    if (d_sample_missing && !g->members[0]->stats.set) return drain(P, fail(g, HPGV_ERR_STATE, "hpgv_set_stats_cohort has not been called"));
    if (d_sample_missing && n_samples > 0) {
        // the counter array is the one destination every member ADDS into: the previous call's adds and its ncclReduce (the
        // OTHER generation's transfers) may still be running when this call's memset is queued.  plan_call made member 0's
        // scan wait for this generation only; with counters, it also waits for the other one.
        GroupMember &M0 = S->m[0];
        DeviceGuard dg(g->members[0]->device);
        const int other = P.gen ^ 1;
        if (M0.xfer_pending[other]) {
            const hipError_t e = hipStreamWaitEvent(M0.scan, M0.xfer_done[other], 0);
            if (e != hipSuccess) return drain(P, fail(g, HPGV_ERR_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(e)));
        }
    }
    for (int k = 0; k < P.G; ++k) {
        hpgv_ctx *mc = g->members[(size_t)k];
        GroupMember &M = S->m[(size_t)k];
        DeviceGuard dg(mc->device);
        int32_t *miss = nullptr;
        if (d_sample_missing && n_samples > 0) {
            // member 0 counts straight into the caller's array, everyone else into a counter array of its own
            if (k == 0) miss = d_sample_missing;
            else {
                const size_t need = (size_t)n_samples * sizeof(int32_t);
                if (M.miss_cap[P.gen] < need) {
                    if (M.d_miss[P.gen]) { (void)hipFree(M.d_miss[P.gen]); M.d_miss[P.gen] = nullptr; M.miss_cap[P.gen] = 0; }
                    HIPCHK(mc, hipMalloc(&M.d_miss[P.gen], need));
                    M.miss_cap[P.gen] = need;
                }
                miss = M.d_miss[P.gen];
            }
            HIPCHK(mc, hipMemsetAsync(miss, 0, (size_t)n_samples * sizeof(int32_t), M.scan));
        }
        if (P.n[k] == 0) continue;
        if (!d_gt[k]) return drain(P, fail(g, HPGV_ERR_INVALID, "member %d has %lld variants but no matrix", k, (long long)P.n[k]));
        const int n = (int)P.n[k];
        int32_t *c8 = (int32_t *)piece_src(P, k, pieces[0], 0);
        rc = hpgv_stats_scan_dev(mc, d_gt[k], n, c8, M.scan);
        if (!rc) rc = hpgv_stats_hwe_dev(mc, c8, n, (double *)piece_src(P, k, pieces[1], 32), (double *)piece_src(P, k, pieces[2], 40), M.scan);
        if (!rc && miss) {
            // k_sample_missing takes at most 65535 bands of rows per launch
            const int step = 65535 * hpgv::SAMPLE_STATS_ROWS;
            const size_t pitch = mc->stats.pitch;
            for (int v0 = 0; v0 < n && !rc; v0 += step)
                rc = hpgv_sample_missing_dev(mc, d_gt[k] + (size_t)v0 * pitch, std::min(step, n - v0), miss, M.scan);
        }
        if (rc) return drain(P, rc);
    }
    rc = exchange(P, pieces);
    if (rc) return drain(P, rc);
    if (!d_sample_missing || n_samples <= 0) return HPGV_OK;
    // the per-sample counters: a context on member 0's device adds its own on ITS transfer stream (which the next call of
    // this generation waits for before it overwrites them), then one ncclReduce (sum, in place on member 0) over the
    // communicator's ranks on member 0's transfer stream, behind those adds and member 0's own scan.
    GroupMember &M0 = S->m[0];
    {
        DeviceGuard dg(g->members[0]->device);
        HIPCHK(g, hipStreamWaitEvent(M0.xfer, M0.scan_done[P.gen], 0));
        for (int k = 1; k < P.G; ++k) {
            GroupMember &M = S->m[(size_t)k];
            if (!M.local) continue;
            HIPCHK(g, hipStreamWaitEvent(M.xfer, M0.scan_done[P.gen], 0));      // member 0's memset of the counters is behind this event
            hipLaunchKernelGGL(k_add_i32, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, M.xfer, d_sample_missing,
                               M.d_miss[P.gen], n_samples);
            HIPCHK(g, hipGetLastError());
            HIPCHK(g, hipEventRecord(M.xfer_done[P.gen], M.xfer));
            HIPCHK(g, hipStreamWaitEvent(M0.xfer, M.xfer_done[P.gen], 0));
        }
    }
    NCCLCHK(g, S, S->GroupStart());
    NCCLCHK(g, S, S->Reduce(d_sample_missing, d_sample_missing, (size_t)n_samples, ncclInt32, ncclSum, 0, S->comms[0], M0.xfer));
    for (int k = 1; k < P.G; ++k) {
        GroupMember &M = S->m[(size_t)k];
        if (M.local) continue;
        NCCLCHK(g, S, S->Reduce(M.d_miss[P.gen], M.d_miss[P.gen], (size_t)n_samples, ncclInt32, ncclSum, 0, S->comms[(size_t)M.rank], M.xfer));
    }
    NCCLCHK(g, S, S->GroupEnd());
    for (int k = 0; k < P.G; ++k) {
        hpgv_ctx *mc = g->members[(size_t)k];
        GroupMember &M = S->m[(size_t)k];
        DeviceGuard dg(mc->device);
        HIPCHK(mc, hipEventRecord(M.xfer_done[P.gen], M.xfer));
        M.xfer_pending[P.gen] = true;
    }
    return HPGV_OK;
    };
    rc = queued();
    return rc ? drain(P, rc) : HPGV_OK;
    HPGV_ABI_CATCH(g)
}

}  // extern "C"

// ---- the epistasis scan over the devices of a group (the reference deals block coordinates to its workers,
//      singlenode/epistasis_runner.c:114-145).  Every combination of `order` SNPs belongs to its FIRST SNP; the first SNPs are
//      cut into G runs of (nearly) equal numbers of combinations -- for pairs at multiples of 64 rows, the tile scan's unit --
//      member g ranks its run on its own device (hpgv_epi_rank_{pairs,triples,order}_rows, one host thread per member), and the
//      ONE exchange is the gather of the members' per-fold top lists (num_folds x max_ranking_size records of 64 bytes) onto
//      member 0 over the group's communicator, where they merge into the whole ranking: a model is in the whole top N only if
//      it is in the top N of its own share.

namespace {

struct EpiRec { double accuracy; int32_t c[5]; int32_t used; uint32_t risky[8]; };     // 64 bytes; used = 0: an empty slot
static_assert(sizeof(EpiRec) == 64, "one record of the gathered top lists");

// combinations of `order` SNPs out of V that begin with one of the first `rows` SNPs
long double epi_combs_before(int V, int order, int rows) {
    auto choose = [](int n, int k) -> long double { if (k < 0 || n < k) return 0.0L; long double r = 1.0L; for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i; return r; };
    return choose(V, order) - choose(V - std::min(rows, V), order);
}

// first SNP where member k's share begins: the first boundary (a multiple of `unit`) with at least k / G of the work before it
int epi_cut(int V, int order, int G, int k, int unit) {
    if (k <= 0) return 0;
    if (k >= G) return V;
    const long double target = epi_combs_before(V, order, V) * k / G;
    int lo = 0, hi = (V + unit - 1) / unit;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (epi_combs_before(V, order, std::min(mid * unit, V)) >= target) hi = mid; else lo = mid + 1;
    }
    return std::min(lo * unit, V);
}

}  // namespace

extern "C" {

int hpgv_group_epi_share(const hpgv_ctx *g, int order, int member, int *i_begin, int *i_end) {
    if (!g || !i_begin || !i_end || order < 2 || order > 5) return HPGV_ERR_INVALID;
    const int G = hpgv_group_size(g);
    if (member < 0 || member >= G) return HPGV_ERR_INVALID;
    const hpgv_ctx *m0 = first_member(g);
    const int V = m0->epi.V, unit = order == 2 ? 64 : 1;
    *i_begin = epi_cut(V, order, G, member, unit);
    *i_end = epi_cut(V, order, G, member + 1, unit);
    return HPGV_OK;
}

int hpgv_group_epi_rank(hpgv_ctx *g, int order, int subset, int max_ranking_size, int32_t *combs_out, double *accuracy,
                        uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms) {
    HPGV_ABI_TRY
    if (!is_group(g)) return fail(g, HPGV_ERR_INVALID, "hpgv_group_epi_rank needs a group context (hpgv_create_multi)");
    if (order < 2 || order > 5) return fail(g, HPGV_ERR_UNSUPPORTED, "combinations of %d SNPs are not supported (2 to 5)", order);
    if (max_ranking_size < 1 || max_ranking_size > 65536 || !combs_out || !accuracy || !risky_mask || !n_ranked)
        return fail(g, HPGV_ERR_INVALID, "bad ranking arguments");
    int rc = hpgv_group_comm_init(g);
    if (rc) return rc;
    GroupState *S = g->grp;
    std::lock_guard<std::mutex> lk(S->mu);
    const int G = (int)g->members.size(), N = max_ranking_size;
    const int nf = g->members[0]->epi.num_folds;
    if (!g->members[0]->epi.have_folds) return fail(g, HPGV_ERR_STATE, "hpgv_epi_set_dataset has not been called");
    for (int k = 1; k < G; ++k)
        if (!g->members[(size_t)k]->epi.have_folds || g->members[(size_t)k]->epi.V != g->members[0]->epi.V || g->members[(size_t)k]->epi.num_folds != nf)
            return fail(g, HPGV_ERR_STATE, "member %d does not hold the dataset and folds of member 0: set them through the group context", k);
    const size_t n_rec = (size_t)nf * (size_t)N, bytes = n_rec * sizeof(EpiRec);
    // ---- every member ranks its share on a host thread of its own ----
    std::vector<std::vector<EpiRec>> lists((size_t)G, std::vector<EpiRec>(n_rec));
    std::vector<int> rcs((size_t)G, HPGV_OK);
    std::vector<float> ms((size_t)G, 0.f);
    {
        std::vector<std::thread> th;
        for (int k = 0; k < G; ++k)
            th.emplace_back([&, k]() {
                try {
                    hpgv_ctx *mc = g->members[(size_t)k];
                    int lo = 0, hi = 0;
                    (void)hpgv_group_epi_share(g, order, k, &lo, &hi);
                    std::vector<int32_t> ci(n_rec), cj(n_rec), ck(n_rec), cn(n_rec * (size_t)order), cnt((size_t)hpgv::EPI_MAX_FOLDS, 0);
                    std::vector<uint32_t> rk(n_rec * 8, 0u);
                    std::vector<double> acc(n_rec, 0.0);
                    int r = HPGV_OK;
                    if (order == 2) r = hpgv_epi_rank_pairs_rows(mc, lo, hi, subset, N, ci.data(), cj.data(), acc.data(), rk.data(), cnt.data(), scan_ms ? &ms[(size_t)k] : nullptr);
                    else if (order == 3) r = hpgv_epi_rank_triples_rows(mc, lo, hi, subset, N, ci.data(), cj.data(), ck.data(), acc.data(), rk.data(), cnt.data(), scan_ms ? &ms[(size_t)k] : nullptr);
                    else r = hpgv_epi_rank_order_rows(mc, order, lo, hi, subset, N, cn.data(), acc.data(), rk.data(), cnt.data(), scan_ms ? &ms[(size_t)k] : nullptr);
                    rcs[(size_t)k] = r;
                    if (r) return;
                    auto &L = lists[(size_t)k];
                    std::memset(L.data(), 0, bytes);
                    for (int f = 0; f < nf; ++f)
                        for (int e = 0; e < cnt[(size_t)f]; ++e) {
                            const size_t o = (size_t)f * (size_t)N + (size_t)e;
                            EpiRec &R = L[o];
                            R.accuracy = acc[o]; R.used = 1;
                            for (int s2 = 0; s2 < 5; ++s2) R.c[s2] = -1;
                            if (order <= 3) { R.c[0] = ci[o]; R.c[1] = cj[o]; if (order == 3) R.c[2] = ck[o]; R.risky[0] = rk[o]; }      // one mask word per model
                            else { for (int s2 = 0; s2 < order; ++s2) R.c[s2] = cn[o * (size_t)order + (size_t)s2]; for (int w = 0; w < 8; ++w) R.risky[w] = rk[o * 8 + (size_t)w]; }
                        }
                } catch (...) { rcs[(size_t)k] = HPGV_ERR_NOMEM; }
            });
        for (auto &t : th) t.join();
    }
    for (int k = 0; k < G; ++k)
        if (rcs[(size_t)k]) return fail(g, rcs[(size_t)k], "member %d: %s", k, hpgv_last_error(g->members[(size_t)k]));
    // ---- the exchange: every member's lists onto member 0 (through the communicator from another device -- and from member 0
    //      itself with the test switch group_self_exchange --, a device-local copy from a context on member 0's device) ----
    struct Bufs { std::vector<void *> p; std::vector<int> dev; ~Bufs() { for (size_t i = 0; i < p.size(); ++i) if (p[i]) { DeviceGuard dg(dev[i]); (void)hipFree(p[i]); } } } bufs;
    auto dev_alloc = [&](int dev, size_t n, void **out) -> hipError_t { DeviceGuard dg(dev); hipError_t e = hipMalloc(out, n); if (e == hipSuccess) { bufs.p.push_back(*out); bufs.dev.push_back(dev); } return e; };
    const int dev0 = g->members[0]->device;
    void *d_all = nullptr;                                            // member 0's device: G lists in member order
    HIPCHK(g, dev_alloc(dev0, bytes * (size_t)G, &d_all));
    std::vector<void *> d_send((size_t)G, nullptr);
    auto by_rccl = [&](int k) { return (k > 0 || g->group_self_exchange) && (!S->m[(size_t)k].local || k == 0); };
    for (int k = 0; k < G; ++k) {
        hpgv_ctx *mc = g->members[(size_t)k];
        GroupMember &M = S->m[(size_t)k];
        DeviceGuard dg(mc->device);
        if (k == 0 && !g->group_self_exchange) { HIPCHK(g, hipMemcpyAsync(d_all, lists[0].data(), bytes, hipMemcpyHostToDevice, M.xfer)); continue; }
        HIPCHK(g, dev_alloc(mc->device, bytes, &d_send[(size_t)k]));
        HIPCHK(g, hipMemcpyAsync(d_send[(size_t)k], lists[(size_t)k].data(), bytes, hipMemcpyHostToDevice, M.xfer));
        if (!by_rccl(k)) HIPCHK(g, hipMemcpyAsync((char *)d_all + bytes * (size_t)k, d_send[(size_t)k], bytes, hipMemcpyDeviceToDevice, M.xfer));
    }
    bool any = false;
    for (int k = 0; k < G; ++k) any = any || by_rccl(k);
    if (any) {
        GroupMember &M0 = S->m[0];
        NCCLCHK(g, S, S->GroupStart());
        for (int k = 0; k < G; ++k) {
            if (!by_rccl(k)) continue;
            GroupMember &M = S->m[(size_t)k];
            NCCLCHK(g, S, S->Send(d_send[(size_t)k], bytes, ncclInt8, 0, S->comms[(size_t)M.rank], M.xfer));
            NCCLCHK(g, S, S->Recv((char *)d_all + bytes * (size_t)k, bytes, ncclInt8, M.rank, S->comms[0], M0.xfer));
        }
        NCCLCHK(g, S, S->GroupEnd());
    }
    for (int k = 0; k < G; ++k) { DeviceGuard dg(g->members[(size_t)k]->device); HIPCHK(g, hipStreamSynchronize(S->m[(size_t)k].xfer)); }
    std::vector<EpiRec> all(n_rec * (size_t)G);
    { DeviceGuard dg(dev0); HIPCHK(g, hipMemcpy(all.data(), d_all, bytes * (size_t)G, hipMemcpyDeviceToHost)); }
    // ---- merge: per fold the best N of the members' lists (add_to_model_ranking, model.c:478-517: higher accuracy, then the
    //      smaller combination) ----
    auto better = [](const EpiRec &a, const EpiRec &b) {
        if (a.accuracy != b.accuracy) return a.accuracy > b.accuracy;
        for (int s2 = 0; s2 < 5; ++s2) if (a.c[s2] != b.c[s2]) return a.c[s2] < b.c[s2];
        return false;
    };
    std::vector<EpiRec> t;
    for (int f = 0; f < nf; ++f) {
        t.clear();
        for (int k = 0; k < G; ++k)
            for (int e = 0; e < N; ++e) { const EpiRec &R = all[(size_t)k * n_rec + (size_t)f * (size_t)N + (size_t)e]; if (R.used) t.push_back(R); }
        std::sort(t.begin(), t.end(), better);
        if ((int)t.size() > N) t.resize((size_t)N);
        n_ranked[f] = (int32_t)t.size();
        for (size_t e = 0; e < t.size(); ++e) {
            const size_t o = (size_t)f * (size_t)N + e;
            for (int s2 = 0; s2 < order; ++s2) combs_out[o * (size_t)order + (size_t)s2] = t[e].c[s2];
            accuracy[o] = t[e].accuracy;
            for (int w = 0; w < 8; ++w) risky_mask[o * 8 + (size_t)w] = t[e].risky[w];
        }
    }
    if (scan_ms) { float m = 0.f; for (float x : ms) m = std::max(m, x); *scan_ms = m; }       // the members scan side by side: the slowest one
    return HPGV_OK;
    HPGV_ABI_CATCH(g)
}

}  // extern "C"
