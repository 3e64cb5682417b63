// group.hip -- jsdr_group_*: ONE process hosting its demodulators on several GPUs (first option).
//
// The reference hosts all its FUNcubeBPSKDemod instances in one JVM (jsdr.java:479-483: a loop of `new FUNcubeBPSKDemod(i, ...)`)
// and one audio thread feeds them all (JavaAudio.java:298-304).  A JVM host has no torch.distributed; what it can call is
// this: the total number of streams is split into contiguous, equal shards (java-sdr_amd/sharding.py: rank r of N owns
// [r S, (r+1) S)), every device gets ONE host thread that owns the device's handles and HIP streams, and after each step
// the per-stream result slots (k_pack_slots: counters, the call's bits, every FECDecode result) of all devices are
// gathered to every device by one ncclAllGather per device (RCCL over xGMI) on that device's gather stream -- beside the
// next step's kernels.  Streams are independent, so there is no other exchange.
//
// librccl.so is loaded with dlopen when the first group is created (the single-GPU library does not depend on it); a
// group that is created with JSDR_GROUP_GATHER_COPY gathers with device-to-device copies instead (what a host without
// RCCL, or a rehearsal with several ranks on ONE device -- RCCL refuses a device twice -- can run).
#include "common.h"
#include <dlfcn.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace jsdr {

// the five RCCL entry points used, resolved at run time (rccl.h:236 ncclCommInitAll, :ncclAllGather, ncclCommDestroy, ncclGetErrorString)
typedef struct ncclComm *ncclComm_t;
struct Rccl {
    void *so = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;  // optional: frees a communicator whose collective will never complete
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
};
static const int NCCL_UINT8 = 1;  // ncclUint8 (rccl.h ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1)

static std::mutex g_rccl_mu;
static Rccl g_rccl;

static int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.so) return JSDR_OK;
    void *so = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (so) break;
    }
    JSDR_REQUIRE(so, "jsdr_group: librccl.so not found (%s); use JSDR_GROUP_GATHER_COPY for a gather without RCCL", dlerror());
    Rccl r;
    r.so = so;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(so, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(so, "ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(so, "ncclCommAbort"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(so, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(so, "ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(so, "ncclGetVersion"));
    JSDR_REQUIRE(r.CommInitAll && r.CommDestroy && r.AllGather && r.GetErrorString, "jsdr_group: librccl.so lacks an entry point");
    g_rccl = r;
    return JSDR_OK;
}

struct Job {
    enum Kind { BATCH, SYNC, CALL, QUIT } kind = BATCH;
    const int16_t *raw = nullptr;
    int64_t stride = 0, nsamples = 0;
    int ic = 0, qc = 0;
    float *psd = nullptr;
    int (*fn)(void *, int) = nullptr;  // CALL: runs on the rank's thread with its device current
    void *arg = nullptr;
};

struct Group;

struct Rank {
    Group *g = nullptr;
    int index = 0, dev = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    // completion of the job most recently posted
    long long posted = 0, done = 0;
    int status = JSDR_OK;
    char err[512] = "";
    jsdr_bpsk *dem = nullptr;
    jsdr_fft *fft = nullptr;
    hipStream_t main = nullptr, psd = nullptr, gather = nullptr;
    uint8_t *slots = nullptr, *gathered = nullptr;
    int64_t slot_bytes = 0;
    hipEvent_t ev_packed = nullptr, ev_pulled = nullptr;
    bool pulled_once = false;
    ncclComm_t comm = nullptr;
};

struct Group {
    int ndev = 0, streams_per_dev = 0, total_streams = 0;
    int rate = 0, frame = 0;
    bool copy_gather = false, with_psd = false;
    int64_t slot_bytes = 0;
    std::vector<Rank *> ranks;
    // a step's two-phase rendezvous: every rank enqueues its kernels and its slot packing, then ALL of them or NONE issue
    // the gather (a rank that failed before the collective must not leave the others waiting in it)
    std::mutex bmu;
    std::condition_variable bcv;
    int arrived = 0, generation = 0;
    bool step_failed = false, step_failed_latched = false;
    long long steps = 0;
    // a collective that failed on ONE rank leaves the others' gather streams waiting for a peer that never joins: from then
    // on nobody waits on a gather stream any more (sync / destroy), the communicators are aborted rather than destroyed, and
    // every later batch call is refused
    std::atomic<bool> broken{false};
};

static void rank_fail(Rank *r, const char *what)
{
    r->status = JSDR_ERR;
    snprintf(r->err, sizeof(r->err), "jsdr_group rank %d (device %d): %s: %s", r->index, r->dev, what, jsdr_last_error());
}

// all ranks meet; returns true when every rank arrived without a failure
static bool step_barrier(Group *g, bool failed)
{
    std::unique_lock<std::mutex> lk(g->bmu);
    if (failed) g->step_failed = true;
    const int gen = g->generation;
    if (++g->arrived == g->ndev) {
        g->arrived = 0;
        g->step_failed_latched = g->step_failed;
        g->step_failed = false;
        g->generation++;
        g->bcv.notify_all();
    } else {
        g->bcv.wait(lk, [&] { return g->generation != gen; });
    }
    return !g->step_failed_latched;
}

static void run_batch(Rank *r, const Job &j)
{
    Group *g = r->g;
    bool failed = false;
    const int S = g->streams_per_dev;
    if (g->broken.load()) {
        r->status = JSDR_ERR;
        snprintf(r->err, sizeof(r->err), "jsdr_group rank %d: the group is broken (an earlier collective failed on some rank); destroy it", r->index);
        step_barrier(g, true);
        return;
    }
    if (g->with_psd && j.psd) {
        // raw is [S][stride]: only when the streams lie back to back (stride == 2 nsamples int16) are their frames ONE run;
        // otherwise (the chunked calls of a longer buffer, the JNI's 2 max_batch stride) the transform goes stream by stream
        const int64_t nf = j.nsamples / g->frame;
        if (j.stride == 2 * j.nsamples) {
            if (jsdr_fft_batch_i16(r->fft, j.raw, (int64_t)S * nf, j.ic, j.qc, j.psd, r->psd) != JSDR_OK) failed = true;
        } else {
            for (int s = 0; s < S && !failed; s++)
                if (jsdr_fft_batch_i16(r->fft, j.raw + (int64_t)s * j.stride, nf, j.ic, j.qc,
                                       j.psd + (int64_t)s * nf * (g->frame + 2), r->psd) != JSDR_OK)
                    failed = true;
        }
        if (failed) rank_fail(r, "jsdr_fft_batch_i16");
    }
    if (!failed && jsdr_bpsk_batch_i16(r->dem, j.raw, j.stride, j.nsamples, j.ic, j.qc, r->main) != JSDR_OK) {
        rank_fail(r, "jsdr_bpsk_batch_i16");
        failed = true;
    }
    if (!failed && g->copy_gather) {
        // the last step's pulls of the other ranks READ this rank's slots: the packing below overwrites them
        for (Rank *o : g->ranks)
            if (o->pulled_once && hipStreamWaitEvent(r->gather, o->ev_pulled, 0) != hipSuccess) {
                set_error("hipStreamWaitEvent failed");
                rank_fail(r, "gather");
                failed = true;
                break;
            }
    }
    // the packing waits (on the gather stream) for the side-stream tail / FEC of this call; the next call's front end on
    // the main stream is not serialised behind it
    if (!failed && jsdr_bpsk_pack_slots(r->dem, r->slots, r->gather) != JSDR_OK) {
        rank_fail(r, "jsdr_bpsk_pack_slots");
        failed = true;
    }
    if (!failed && g->copy_gather && hipEventRecord(r->ev_packed, r->gather) != hipSuccess) {
        set_error("hipEventRecord failed");
        rank_fail(r, "gather");
        failed = true;
    }
    const bool go = step_barrier(g, failed);
    if (!go) {
        if (!failed) {
            r->status = JSDR_ERR;
            snprintf(r->err, sizeof(r->err), "jsdr_group rank %d: another rank failed before the gather; step abandoned", r->index);
        }
        return;
    }
    const size_t seg = (size_t)S * (size_t)g->slot_bytes;
    if (!g->copy_gather) {
        // Every device thread calls ncclAllGather on ITS OWN communicator, without ncclGroupStart / ncclGroupEnd.  That is the
        // form rccl.h prescribes for this layout ("Collective communication operations must be called separately for each
        // communicator in a communicator clique ... each call has to be done from a different thread or process, or need to
        // use Group Semantics", rccl.h:522-529; the group calls are for "managing multiple GPUs from a single thread", :899) --
        // one thread per communicator is the first alternative, the rendezvous above guarantees every thread gets here.
        const int rc = g_rccl.AllGather(r->slots, r->gathered, seg, NCCL_UINT8, r->comm, r->gather);
        if (rc != 0) {
            r->status = JSDR_ERR;
            snprintf(r->err, sizeof(r->err), "jsdr_group rank %d: ncclAllGather: %s", r->index, g_rccl.GetErrorString(rc));
        }
        // a collective that one rank could not enqueue never completes on the others: everybody learns of it HERE, before
        // anyone waits on a gather stream
        if (!step_barrier(g, rc != 0)) g->broken.store(true);
    } else {
        // every rank PULLS every rank's segment into its own gathered buffer on its own gather stream, after that rank's
        // packing (event recorded before the rendezvous above)
        for (Rank *o : g->ranks) {
            hipError_t e = hipStreamWaitEvent(r->gather, o->ev_packed, 0);
            if (e == hipSuccess)
                e = hipMemcpyAsync(r->gathered + (size_t)o->index * seg, o->slots, seg, hipMemcpyDeviceToDevice, r->gather);
            if (e != hipSuccess) {
                r->status = JSDR_ERR;
                snprintf(r->err, sizeof(r->err), "jsdr_group rank %d: copy gather from rank %d: %s", r->index, o->index, hipGetErrorString(e));
                break;
            }
        }
        (void)hipEventRecord(r->ev_pulled, r->gather);
        // the others' slots are read by those copies: nobody re-packs before every rank has ENQUEUED its pulls (rendezvous),
        // and the next packing waits for ev_pulled of every rank (above)
        step_barrier(g, false);
        r->pulled_once = true;
    }
}

static void run_sync(Rank *r)
{
    if (r->fft && hipStreamSynchronize(r->psd) != hipSuccess) {
        set_error("hipStreamSynchronize(psd) failed");
        rank_fail(r, "sync");
    }
    if (hipStreamSynchronize(r->main) != hipSuccess) {
        set_error("hipStreamSynchronize(main) failed");
        rank_fail(r, "sync");
    }
    if (jsdr_bpsk_sync(r->dem) != JSDR_OK) rank_fail(r, "jsdr_bpsk_sync");
    if (r->g->broken.load()) {
        set_error("the group is broken: a collective failed on some rank, the gather streams are not waited for");
        rank_fail(r, "sync");
    } else if (hipStreamSynchronize(r->gather) != hipSuccess) {
        set_error("hipStreamSynchronize(gather) failed");
        rank_fail(r, "sync");
    }
}

static void rank_thread(Rank *r)
{
    (void)hipSetDevice(r->dev);
    for (;;) {
        Job j;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv.wait(lk, [&] { return !r->q.empty(); });
            j = r->q.front();
            r->q.pop_front();
        }
        if (j.kind == Job::QUIT) break;
        if (j.kind == Job::BATCH) run_batch(r, j);
        else if (j.kind == Job::SYNC) run_sync(r);
        else if (j.kind == Job::CALL) {
            if (j.fn(j.arg, r->index) != JSDR_OK) rank_fail(r, "call");
        }
        {
            std::lock_guard<std::mutex> lk(r->mu);
            r->done++;
        }
        r->cv.notify_all();
    }
}

static void post(Rank *r, const Job &j)
{
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->q.push_back(j);
        r->posted++;
    }
    r->cv.notify_all();
}

static void wait_done(Rank *r)
{
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv.wait(lk, [&] { return r->done == r->posted; });
}

// post one job to every rank, wait until every rank's thread has finished it (for BATCH: has ENQUEUED its work on the
// device); the first rank's error, if any, becomes the caller's jsdr_last_error()
static int post_all(Group *g, std::vector<Job> &jobs)
{
    for (int i = 0; i < g->ndev; i++) post(g->ranks[i], jobs[i]);
    for (Rank *r : g->ranks) wait_done(r);
    // the rank that really failed speaks, not one that only abandoned the step because of it; every status is cleared
    // (reported; a later call may succeed -- the handles keep their own state)
    Rank *bad = nullptr;
    for (Rank *r : g->ranks)
        if (r->status != JSDR_OK && (!bad || (strstr(bad->err, "another rank failed") && !strstr(r->err, "another rank failed")))) bad = r;
    for (Rank *r : g->ranks) r->status = JSDR_OK;
    if (bad) {
        set_error("%s", bad->err);
        return JSDR_ERR;
    }
    return JSDR_OK;
}

struct CreateArgs {
    Group *g;
    int rate, frame, tuning, do_fft, do_up;
    int64_t max_batch;
};

static int rank_create(void *p, int index)
{
    CreateArgs *a = static_cast<CreateArgs *>(p);
    Group *g = a->g;
    Rank *r = g->ranks[index];
    JSDR_HIP_TRY(hipStreamCreateWithFlags(&r->main, hipStreamNonBlocking));
    JSDR_HIP_TRY(hipStreamCreateWithFlags(&r->gather, hipStreamNonBlocking));
    JSDR_HIP_TRY(hipEventCreateWithFlags(&r->ev_packed, hipEventDisableTiming));
    JSDR_HIP_TRY(hipEventCreateWithFlags(&r->ev_pulled, hipEventDisableTiming));
    if (jsdr_bpsk_create(&r->dem, a->rate, a->frame, a->tuning, a->do_fft, a->do_up, g->streams_per_dev, a->max_batch) != JSDR_OK)
        return JSDR_ERR;
    if (g->with_psd) {
        JSDR_HIP_TRY(hipStreamCreateWithFlags(&r->psd, hipStreamNonBlocking));
        if (jsdr_fft_create(&r->fft, a->frame, a->rate) != JSDR_OK) return JSDR_ERR;
    }
    // fft.receive and FUNcubeBPSKDemod.receive over the same batch: where the library says the CU split pays (jsdr_bpsk_pair_shares:
    // from 8192 streams per device) the two batch kernels run SIDE BY SIDE on the device's PSD and main streams
    if (g->with_psd) {
        int sf = 0, sd = 0;
        if (jsdr_bpsk_pair_shares(r->dem, &sf, &sd) != JSDR_OK || jsdr_fft_set_cu_share(r->fft, sf) != JSDR_OK ||
            jsdr_bpsk_set_cu_share(r->dem, sd) != JSDR_OK)
            return JSDR_ERR;
    }
    int64_t sb = 0;
    if (jsdr_bpsk_slot_info(r->dem, &sb, nullptr, nullptr, nullptr, nullptr) != JSDR_OK) return JSDR_ERR;
    r->slot_bytes = sb;  // (the creating thread copies rank 0's into the group once every rank is done: no shared write here)
    const size_t seg = (size_t)g->streams_per_dev * (size_t)sb;
    JSDR_HIP_TRY(hipMalloc((void **)&r->slots, seg > 64 ? seg : 64));
    const size_t gbytes = (seg > 64 ? seg : 64) * (size_t)g->ndev;
    JSDR_HIP_TRY(hipMalloc((void **)&r->gathered, gbytes));
    JSDR_HIP_TRY(hipMemset(r->gathered, 0, gbytes));
    if (g->copy_gather) {
        for (Rank *o : g->ranks)
            if (o->dev != r->dev) {
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, r->dev, o->dev);
                if (can) {
                    hipError_t e = hipDeviceEnablePeerAccess(o->dev, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
            }
    }
    return JSDR_OK;
}

// Create-time self-test of the gather path: every device contributes 64 bytes that name its rank, one all-gather over the
// communicators (or the copy gather's pulls) moves them, and every device checks every segment -- so that a machine whose
// RCCL cannot connect its devices fails HERE, with a message that says so, and not in the first step's slot check.
static int rank_selftest(void *p, int index)
{
    Group *g = static_cast<Group *>(p);
    Rank *r = g->ranks[index];
    const size_t seg = 64;
    uint8_t mine[64];
    for (int i = 0; i < 64; i++) mine[i] = (uint8_t)(0xA5 ^ (index * 37 + i));
    bool failed = false;
    if (hipMemcpy(r->slots, mine, seg, hipMemcpyHostToDevice) != hipSuccess) failed = true;
    if (!failed && g->copy_gather && hipEventRecord(r->ev_packed, r->gather) != hipSuccess) failed = true;
    if (failed) {
        set_error("copying the test pattern to the device failed");
        rank_fail(r, "self-test");
    }
    if (!step_barrier(g, failed)) {
        if (!failed) {
            r->status = JSDR_ERR;
            snprintf(r->err, sizeof(r->err), "jsdr_group rank %d: another rank failed before the gather; step abandoned", r->index);
        }
        return r->status;
    }
    if (!g->copy_gather) {
        const int rc = g_rccl.AllGather(r->slots, r->gathered, seg, NCCL_UINT8, r->comm, r->gather);
        if (rc != 0) {
            set_error("ncclAllGather: %s", g_rccl.GetErrorString(rc));
            rank_fail(r, "self-test");
        }
        if (!step_barrier(g, rc != 0)) {  // (see run_batch: nobody waits for a collective some rank did not enqueue)
            g->broken.store(true);
            if (r->status == JSDR_OK) {
                r->status = JSDR_ERR;
                snprintf(r->err, sizeof(r->err), "jsdr_group rank %d: another rank failed before the gather; step abandoned", r->index);
            }
            return JSDR_ERR;
        }
    } else {
        for (Rank *o : g->ranks) {
            hipError_t e = hipStreamWaitEvent(r->gather, o->ev_packed, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(r->gathered + (size_t)o->index * seg, o->slots, seg, hipMemcpyDeviceToDevice, r->gather);
            if (e != hipSuccess) {
                set_error("copy gather from rank %d: %s", o->index, hipGetErrorString(e));
                rank_fail(r, "self-test");
                break;
            }
        }
    }
    std::vector<uint8_t> got(seg * (size_t)g->ndev);
    bool ok = r->status == JSDR_OK && hipStreamSynchronize(r->gather) == hipSuccess &&
              hipMemcpy(got.data(), r->gathered, got.size(), hipMemcpyDeviceToHost) == hipSuccess;
    if (g->copy_gather) step_barrier(g, false);  // nobody overwrites its 64 bytes while another rank may still pull them
    if (r->status != JSDR_OK) return JSDR_ERR;
    if (!ok) {
        set_error("waiting for / reading back the gathered test pattern failed");
        rank_fail(r, "self-test");
        return JSDR_ERR;
    }
    for (int o = 0; o < g->ndev; o++)
        for (int i = 0; i < 64; i++)
            if (got[(size_t)o * seg + i] != (uint8_t)(0xA5 ^ (o * 37 + i))) {
                set_error("device index %d received byte %d of rank %d's segment as 0x%02x, expected 0x%02x: the gather between the devices does not work",
                          index, i, o, got[(size_t)o * seg + i], (uint8_t)(0xA5 ^ (o * 37 + i)));
                rank_fail(r, "self-test");
                return JSDR_ERR;
            }
    (void)hipMemset(r->gathered, 0, got.size());
    return JSDR_OK;
}

static void group_free(Group *g)
{
    for (Rank *r : g->ranks) {
        if (r->th.joinable()) {
            Job q;
            q.kind = Job::QUIT;
            post(r, q);
            r->th.join();
        }
    }
    // nothing of any device may still be in flight when a communicator, a handle or a buffer goes away
    for (Rank *r : g->ranks) {
        (void)hipSetDevice(r->dev);
        if (r->gather && !g->broken.load()) (void)hipStreamSynchronize(r->gather);
        if (r->main) (void)hipStreamSynchronize(r->main);
        if (r->psd) (void)hipStreamSynchronize(r->psd);
    }
    for (Rank *r : g->ranks) {
        (void)hipSetDevice(r->dev);
        if (r->comm && g->broken.load() && g_rccl.CommAbort) (void)g_rccl.CommAbort(r->comm);
        else if (r->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(r->comm);
        if (r->dem) (void)jsdr_bpsk_destroy(r->dem);
        if (r->fft) (void)jsdr_fft_destroy(r->fft);
        if (r->slots) (void)hipFree(r->slots);
        if (r->gathered) (void)hipFree(r->gathered);
        if (r->ev_packed) (void)hipEventDestroy(r->ev_packed);
        if (r->ev_pulled) (void)hipEventDestroy(r->ev_pulled);
        if (r->main) (void)hipStreamDestroy(r->main);
        if (r->psd) (void)hipStreamDestroy(r->psd);
        if (r->gather) (void)hipStreamDestroy(r->gather);
        delete r;
    }
    delete g;
}

}  // namespace jsdr

using namespace jsdr;

struct jsdr_group {
    Group *g;
};

extern "C" {

int jsdr_group_create(jsdr_group **out, int ndev, const int *devices, int rate, int nsamples_per_frame, int tuning_hz, int do_fft,
                      int do_up, int total_streams, int64_t max_batch_samples, int flags)
{
    JSDR_REQUIRE(out, "jsdr_group_create: null handle pointer");
    *out = nullptr;
    JSDR_REQUIRE(ndev >= 1 && ndev <= 64, "jsdr_group_create: ndev = %d", ndev);
    JSDR_REQUIRE(total_streams >= ndev && total_streams % ndev == 0,
                 "jsdr_group_create: %d streams do not split evenly over %d devices (the gather needs equal counts)", total_streams, ndev);
    int have = 0;
    JSDR_HIP_TRY(hipGetDeviceCount(&have));
    const bool copy_gather = (flags & JSDR_GROUP_GATHER_COPY) != 0;
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) {
        devs[i] = devices ? devices[i] : i;
        JSDR_REQUIRE(devs[i] >= 0 && devs[i] < have, "jsdr_group_create: device %d of %d visible", devs[i], have);
        for (int k = 0; k < i; k++)
            JSDR_REQUIRE(copy_gather || devs[k] != devs[i],
                         "jsdr_group_create: device %d listed twice (RCCL takes a device once; JSDR_GROUP_GATHER_COPY allows it)", devs[i]);
    }
    if (!copy_gather && rccl_load() != JSDR_OK) return JSDR_ERR;
    int prev = 0;
    JSDR_HIP_TRY(hipGetDevice(&prev));
    Group *g = new Group();
    g->ndev = ndev;
    g->total_streams = total_streams;
    g->streams_per_dev = total_streams / ndev;
    g->rate = rate;
    g->frame = nsamples_per_frame;
    g->copy_gather = copy_gather;
    g->with_psd = (flags & JSDR_GROUP_WITH_PSD) != 0;
    for (int i = 0; i < ndev; i++) {
        Rank *r = new Rank();
        r->g = g;
        r->index = i;
        r->dev = devs[i];
        g->ranks.push_back(r);
    }
    for (Rank *r : g->ranks) r->th = std::thread(rank_thread, r);
    CreateArgs ca{g, rate, nsamples_per_frame, tuning_hz, do_fft, do_up, max_batch_samples};
    std::vector<Job> jobs(ndev);
    for (auto &j : jobs) {
        j.kind = Job::CALL;
        j.fn = rank_create;
        j.arg = &ca;
    }
    if (post_all(g, jobs) != JSDR_OK) {
        group_free(g);
        (void)hipSetDevice(prev);
        return JSDR_ERR;
    }
    if (!copy_gather) {
        // one communicator per device, created together (rccl.h:236); the calling thread's device is restored afterwards
        std::vector<ncclComm_t> comms(ndev, nullptr);
        const int rc = g_rccl.CommInitAll(comms.data(), ndev, devs.data());
        (void)hipSetDevice(prev);
        if (rc != 0) {
            set_error("jsdr_group_create: ncclCommInitAll over %d device(s): %s", ndev, g_rccl.GetErrorString(rc));
            group_free(g);
            return JSDR_ERR;
        }
        for (int i = 0; i < ndev; i++) g->ranks[i]->comm = comms[i];
    }
    (void)hipSetDevice(prev);
    g->slot_bytes = g->ranks[0]->slot_bytes;
    for (Rank *r : g->ranks)
        if (r->slot_bytes != g->slot_bytes) {
            set_error("jsdr_group_create: rank %d packs %lld-byte slots, rank 0 %lld", r->index, (long long)r->slot_bytes, (long long)g->slot_bytes);
            group_free(g);
            return JSDR_ERR;
        }
    // the gather path, exercised once before the first step (64 bytes per device)
    for (auto &j : jobs) {
        j.kind = Job::CALL;
        j.fn = rank_selftest;
        j.arg = g;
    }
    if (post_all(g, jobs) != JSDR_OK) {
        const std::string why = jsdr_last_error();
        group_free(g);
        (void)hipSetDevice(prev);
        set_error("jsdr_group_create: gather self-test failed: %s", why.c_str());
        return JSDR_ERR;
    }
    (void)hipSetDevice(prev);
    jsdr_group *h = new jsdr_group();
    h->g = g;
    *out = h;
    return JSDR_OK;
}

int jsdr_group_destroy(jsdr_group *h)
{
    if (!h) return JSDR_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    group_free(h->g);
    (void)hipSetDevice(prev);
    delete h;
    return JSDR_OK;
}

int jsdr_group_info(jsdr_group *h, int *ndev, int *streams_per_device, int64_t *slot_bytes, int *rccl_version)
{
    JSDR_REQUIRE(h, "jsdr_group_info: null handle");
    if (ndev) *ndev = h->g->ndev;
    if (streams_per_device) *streams_per_device = h->g->streams_per_dev;
    if (slot_bytes) *slot_bytes = h->g->slot_bytes;
    if (rccl_version) {
        *rccl_version = 0;
        if (!h->g->copy_gather && g_rccl.GetVersion) (void)g_rccl.GetVersion(rccl_version);
    }
    return JSDR_OK;
}

int jsdr_group_device(jsdr_group *h, int index, int *device, jsdr_bpsk **dem, jsdr_fft **fft)
{
    JSDR_REQUIRE(h && index >= 0 && index < h->g->ndev, "jsdr_group_device: bad argument");
    Rank *r = h->g->ranks[index];
    if (device) *device = r->dev;
    if (dem) *dem = r->dem;
    if (fft) *fft = r->fft;
    return JSDR_OK;
}

int jsdr_group_batch_i16(jsdr_group *h, const int16_t *const *raw_dev, int64_t stream_stride_i16, int64_t nsamples, int ic, int qc,
                         float *const *psd_dev)
{
    JSDR_REQUIRE(h && raw_dev, "jsdr_group_batch_i16: null argument");
    Group *g = h->g;
    JSDR_REQUIRE(nsamples >= 0, "jsdr_group_batch_i16: negative sample count");
    JSDR_REQUIRE(!psd_dev || g->with_psd, "jsdr_group_batch_i16: PSD buffers given to a group created without JSDR_GROUP_WITH_PSD");
    JSDR_REQUIRE(!psd_dev || nsamples % g->frame == 0, "jsdr_group_batch_i16: the PSD needs whole frames (%d samples)", g->frame);
    std::vector<Job> jobs(g->ndev);
    for (int i = 0; i < g->ndev; i++) {
        JSDR_REQUIRE(raw_dev[i], "jsdr_group_batch_i16: null input for device index %d", i);
        jobs[i].kind = Job::BATCH;
        jobs[i].raw = raw_dev[i];
        jobs[i].stride = stream_stride_i16;
        jobs[i].nsamples = nsamples;
        jobs[i].ic = ic;
        jobs[i].qc = qc;
        jobs[i].psd = psd_dev ? psd_dev[i] : nullptr;
    }
    g->steps++;
    return post_all(g, jobs);
}

int jsdr_group_sync(jsdr_group *h)
{
    JSDR_REQUIRE(h, "jsdr_group_sync: null handle");
    std::vector<Job> jobs(h->g->ndev);
    for (auto &j : jobs) j.kind = Job::SYNC;
    return post_all(h->g, jobs);
}

int jsdr_group_gathered(jsdr_group *h, int index, const uint8_t **slots_dev, int64_t *bytes)
{
    JSDR_REQUIRE(h && index >= 0 && index < h->g->ndev && slots_dev, "jsdr_group_gathered: bad argument");
    *slots_dev = h->g->ranks[index]->gathered;
    if (bytes) *bytes = (int64_t)h->g->total_streams * h->g->slot_bytes;
    return JSDR_OK;
}

int jsdr_group_read_slot(jsdr_group *h, int index, int stream, uint8_t *slot_host)
{
    JSDR_REQUIRE(h && index >= 0 && index < h->g->ndev && slot_host, "jsdr_group_read_slot: bad argument");
    Group *g = h->g;
    JSDR_REQUIRE(stream >= 0 && stream < g->total_streams, "jsdr_group_read_slot: stream %d of %d", stream, g->total_streams);
    if (jsdr_group_sync(h) != JSDR_OK) return JSDR_ERR;
    int prev = 0;
    JSDR_HIP_TRY(hipGetDevice(&prev));
    Rank *r = g->ranks[index];
    JSDR_HIP_TRY(hipSetDevice(r->dev));
    const hipError_t e = hipMemcpy(slot_host, r->gathered + (size_t)stream * (size_t)g->slot_bytes, (size_t)g->slot_bytes, hipMemcpyDeviceToHost);
    (void)hipSetDevice(prev);
    JSDR_HIP_TRY(e);
    return JSDR_OK;
}

}  // extern "C"
