// Launch plans: the host side of one frame, recorded once and replayed from C.
//
// The reference's forward is a Python loop over 24 blocks issuing ~100 torch ops each (toc3d_eva_vit.py:263-291); the first
// version of this library kept that shape (Python sequencing ~230 C-ABI calls per frame) and was host-bound at 4.6 ms of a
// 5.4 ms frame.  All shapes, buffers and launch parameters of a frame are static per (config, input shape), so the sequence is
// recorded ONCE (toc3d_plan_begin .. toc3d_plan_end: every toc3d_* call made by the recording thread stores its kernel launches
// here instead of running them) and then replayed by toc3d_plan_run:
//   mode 0: hipLaunchKernel on one HIP stream per *lane* (lane 0 = the caller's stream), cross-lane edges as hipEvent record / wait;
//   mode 1: one explicitly constructed hipGraph (hipGraphAddKernelNode with the lane order + cross-lane edges as node dependencies;
//           no stream capture), launched on the caller's stream.
// Lanes are how the recording names concurrency: the `stream` argument of a recorded call is a lane handle
// (toc3d_plan_lane_stream), and toc3d_plan_wait(plan, a, b) orders lane a's next launch behind everything recorded so far on lane b.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "capi.h"

thread_local Toc3dPlan* toc3d_tls_recording = nullptr;

namespace {
constexpr intptr_t LANE_TAG = 0x70c3d000;        // lane handle = LANE_TAG + lane (never a valid hipStream_t: not 16-byte aligned heap memory)
constexpr int MAX_LANES = 64;                 // toc3d_amd/plan.py MAX_LANES mirrors this (frames with more lanes launch eagerly)
}  // namespace

struct Toc3dPlan {
    struct Node {
        const void* func;
        dim3 grid, block;
        uint32_t lds;
        int lane;
        uint32_t first_arg, nargs;               // into arg_off
        std::vector<int> deps;                   // nodes of other lanes that must finish first
        int signal = -1;                         // event to record behind this node (mode 0), -1 = none
        std::vector<int> waits;                  // events to wait for in front of this node (mode 0)
    };
    std::vector<Node> nodes;
    std::vector<uint32_t> arg_off;               // byte offset of every argument value in blob
    std::vector<char> blob;
    int last_on_lane[MAX_LANES];
    std::vector<int> pending[MAX_LANES];         // cross-lane dependencies the next node of the lane inherits
    bool lane_used[MAX_LANES] = {};
    bool recording = false, finalized = false;
    int mode = 0, device = -1;
    // replay state
    std::vector<hipStream_t> streams;            // lanes 1.. (shared by every plan of the process: shared_lane_stream below; never destroyed)
    std::vector<hipEvent_t> events;              // one per signalling node (owned)
    hipEvent_t entry = nullptr;
    std::vector<hipEvent_t> lane_done;           // lanes 1..: joined into lane 0 at the end of a run
    std::vector<void*> argv;                     // per node: pointers into blob
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;

    Toc3dPlan() {
        for (int& l : last_on_lane) l = -1;
    }
    void release() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        for (hipEvent_t e : lane_done) (void)hipEventDestroy(e);
        if (entry) (void)hipEventDestroy(entry);
        exec = nullptr; graph = nullptr; entry = nullptr;
        streams.clear(); events.clear(); lane_done.clear();
    }
};

void toc3d_plan_record(Toc3dPlan* p, const void* func, dim3 grid, dim3 block, size_t lds, hipStream_t lane_handle,
                       const void* const* arg_ptrs, const size_t* arg_sizes, const size_t* arg_aligns, int nargs) {
    intptr_t lane = reinterpret_cast<intptr_t>(lane_handle) - LANE_TAG;
    if (lane < 0 || lane >= MAX_LANES) {
        // a real stream (or NULL) was passed while recording: keep the plan usable and flag the error at toc3d_plan_end
        toc3d_set_error("toc3d_plan: a call made while recording did not pass a lane handle as its stream (toc3d_plan_lane_stream)");
        p->mode = -1;
        lane = 0;
    }
    Toc3dPlan::Node n;
    n.func = func; n.grid = grid; n.block = block; n.lds = (uint32_t)lds; n.lane = (int)lane;
    n.first_arg = (uint32_t)p->arg_off.size(); n.nargs = (uint32_t)nargs;
    for (int i = 0; i < nargs; ++i) {
        size_t off = (p->blob.size() + arg_aligns[i] - 1) / arg_aligns[i] * arg_aligns[i];
        off = (off + 15) / 16 * 16;              // every value 16-byte aligned inside the blob
        p->blob.resize(off + arg_sizes[i]);
        std::memcpy(p->blob.data() + off, arg_ptrs[i], arg_sizes[i]);
        p->arg_off.push_back((uint32_t)off);
    }
    n.deps.swap(p->pending[lane]);
    p->pending[lane].clear();
    p->last_on_lane[lane] = (int)p->nodes.size();
    p->lane_used[lane] = true;
    p->nodes.push_back(std::move(n));
}

namespace {

#define PLAN_HIP(call)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            toc3d_set_error("toc3d_plan: %s failed: %s", #call, hipGetErrorString(e_));       \
            return TOC3D_ERR_LAUNCH;                                                          \
        }                                                                                     \
    } while (0)

// Flags of the events behind the cross-lane edges: HIP's default (system-scope fence at the record).  The events are only ever waited on by streams of the same
// device, so hipEventReleaseToDevice / hipEventDisableSystemFence would be legal -- both were measured 0.4-0.7 % SLOWER (profiles/r03_event_fence.txt: the gap behind
// a signalling launch is the command processor's barrier packet, not the cache write-back) and are not offered.
unsigned event_flags() { return (unsigned)hipEventDisableTiming; }

// The side lanes of EVERY plan of the process run on one set of HIP streams per device, created by the first plan that needs them.  Round 4 finding
// (profiles/r04_stream_priority.txt, tools/ubench/schedule_ab.py with identical schedules): with private streams per plan, the process's first model ran
// 2.3 % faster than every identical model built later -- HIP deals streams round-robin onto a few hardware queues, and a later plan's side lanes can land
// on the queue of the caller's stream, which serialises the scorer's side work behind the block chain.  Sharing the first plan's streams gives every plan
// its mapping.  Correctness never depended on the streams being private (cross-lane order is carried by events; two plans replayed concurrently from two
// host threads merely share the side queues).  Immutable once created: the "no mutable global state" rule of include/toc3d.h is about data, and holds.
int shared_lane_stream(int lane, hipStream_t* out) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, hipStream_t> pool;
    int dev = 0;
    PLAN_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = pool.find({dev, lane});
    if (it == pool.end()) {
        hipStream_t s;
        PLAN_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        it = pool.emplace(std::make_pair(dev, lane), s).first;
    }
    *out = it->second;
    return TOC3D_OK;
}

int build_streams(Toc3dPlan* p) {
    int nlanes = 1;
    for (int l = 1; l < MAX_LANES; ++l)
        if (p->lane_used[l]) nlanes = l + 1;
    for (int l = 1; l < nlanes; ++l) {
        hipStream_t s;
        // (round 4, measured and not taken: side lanes created with hipStreamCreateWithPriority, lowest or highest -- 207.6 / 205.2 frames/s against 203.5 in
        // one hardware-queue mapping, 101 / 81 frames/s in another: profiles/r04_stream_priority.txt)
        { const int rc = shared_lane_stream(l, &s); if (rc != TOC3D_OK) return rc; }
        p->streams.push_back(s);
        hipEvent_t e;
        PLAN_HIP(hipEventCreateWithFlags(&e, event_flags()));
        p->lane_done.push_back(e);
    }
    PLAN_HIP(hipEventCreateWithFlags(&p->entry, event_flags()));
    // one event per node that some other lane waits on
    for (size_t i = 0; i < p->nodes.size(); ++i)
        for (int d : p->nodes[i].deps) {
            Toc3dPlan::Node& src = p->nodes[d];
            if (src.signal < 0) {
                hipEvent_t e;
                PLAN_HIP(hipEventCreateWithFlags(&e, event_flags()));
                src.signal = (int)p->events.size();
                p->events.push_back(e);
            }
            p->nodes[i].waits.push_back(src.signal);
        }
    return TOC3D_OK;
}

int build_graph(Toc3dPlan* p) {
    PLAN_HIP(hipGraphCreate(&p->graph, 0));
    std::vector<hipGraphNode_t> gn(p->nodes.size());
    int last[MAX_LANES];
    for (int& l : last) l = -1;
    for (size_t i = 0; i < p->nodes.size(); ++i) {
        const Toc3dPlan::Node& n = p->nodes[i];
        hipKernelNodeParams kp;
        std::memset(&kp, 0, sizeof(kp));
        kp.func = const_cast<void*>(n.func);
        kp.gridDim = n.grid;
        kp.blockDim = n.block;
        kp.sharedMemBytes = n.lds;
        kp.kernelParams = p->argv.data() + n.first_arg;
        kp.extra = nullptr;
        std::vector<hipGraphNode_t> deps;
        if (last[n.lane] >= 0) deps.push_back(gn[last[n.lane]]);
        for (int d : n.deps) deps.push_back(gn[d]);
        PLAN_HIP(hipGraphAddKernelNode(&gn[i], p->graph, deps.data(), deps.size(), &kp));
        last[n.lane] = (int)i;
    }
    PLAN_HIP(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
    return TOC3D_OK;
}

}  // namespace

extern "C" {

int toc3d_plan_create(toc3d_plan_t* out) {
    TOC3D_REQUIRE(out, "toc3d_plan_create: null output");
    *out = reinterpret_cast<toc3d_plan_t>(new Toc3dPlan());
    return TOC3D_OK;
}

int toc3d_plan_destroy(toc3d_plan_t plan) {
    Toc3dPlan* p = reinterpret_cast<Toc3dPlan*>(plan);
    if (!p) return TOC3D_OK;
    if (toc3d_tls_recording == p) toc3d_tls_recording = nullptr;
    p->release();
    delete p;
    return TOC3D_OK;
}

toc3d_stream_t toc3d_plan_lane_stream(int64_t lane) {
    if (lane < 0 || lane >= MAX_LANES) return nullptr;
    return reinterpret_cast<toc3d_stream_t>(LANE_TAG + (intptr_t)lane);
}

int toc3d_plan_begin(toc3d_plan_t plan) {
    Toc3dPlan* p = reinterpret_cast<Toc3dPlan*>(plan);
    TOC3D_REQUIRE(p, "toc3d_plan_begin: null plan");
    TOC3D_REQUIRE(!toc3d_tls_recording, "toc3d_plan_begin: this thread is already recording a plan");
    TOC3D_REQUIRE(!p->finalized && p->nodes.empty(), "toc3d_plan_begin: plan already holds a recording");
    p->recording = true;
    toc3d_tls_recording = p;
    return TOC3D_OK;
}

int toc3d_plan_wait(toc3d_plan_t plan, int64_t waiting_lane, int64_t on_lane) {
    Toc3dPlan* p = reinterpret_cast<Toc3dPlan*>(plan);
    TOC3D_REQUIRE(p && p->recording, "toc3d_plan_wait: plan is not recording");
    TOC3D_REQUIRE(waiting_lane >= 0 && waiting_lane < MAX_LANES && on_lane >= 0 && on_lane < MAX_LANES, "toc3d_plan_wait: bad lane");
    if (waiting_lane == on_lane) return TOC3D_OK;
    p->lane_used[waiting_lane] = true;
    // everything recorded so far on `on_lane` (its last node, and transitively what that node waited for) plus whatever `on_lane` itself
    // is still going to inherit: an empty lane that was told to wait passes that obligation on
    if (p->last_on_lane[on_lane] >= 0) p->pending[waiting_lane].push_back(p->last_on_lane[on_lane]);
    for (int d : p->pending[on_lane]) p->pending[waiting_lane].push_back(d);
    return TOC3D_OK;
}

int toc3d_plan_end(toc3d_plan_t plan, int mode) {
    Toc3dPlan* p = reinterpret_cast<Toc3dPlan*>(plan);
    TOC3D_REQUIRE(p && p->recording && toc3d_tls_recording == p, "toc3d_plan_end: plan is not being recorded by this thread");
    toc3d_tls_recording = nullptr;
    p->recording = false;
    if (p->mode == -1) return TOC3D_ERR_ARG;     // message set by toc3d_plan_record
    TOC3D_REQUIRE(mode == 0 || mode == 1, "toc3d_plan_end: mode must be 0 (streams) or 1 (hipGraph)");
    TOC3D_REQUIRE(!p->nodes.empty(), "toc3d_plan_end: nothing was recorded");
    p->mode = mode;
    PLAN_HIP(hipGetDevice(&p->device));
    p->argv.resize(p->arg_off.size());
    for (size_t i = 0; i < p->arg_off.size(); ++i) p->argv[i] = p->blob.data() + p->arg_off[i];
    const int rc = mode == 0 ? build_streams(p) : build_graph(p);
    if (rc != TOC3D_OK) { p->release(); return rc; }
    p->finalized = true;
    return TOC3D_OK;
}

int64_t toc3d_plan_num_launches(toc3d_plan_t plan) {
    Toc3dPlan* p = reinterpret_cast<Toc3dPlan*>(plan);
    return p ? (int64_t)p->nodes.size() : -1;
}

int toc3d_plan_run(toc3d_plan_t plan, toc3d_stream_t stream) {
    Toc3dPlan* p = reinterpret_cast<Toc3dPlan*>(plan);
    TOC3D_REQUIRE(p && p->finalized, "toc3d_plan_run: plan is not finalized (toc3d_plan_end)");
    hipStream_t s0 = as_stream(stream);
    if (p->mode == 1) {
        PLAN_HIP(hipGraphLaunch(p->exec, s0));
        return TOC3D_OK;
    }
    const size_t nside = p->streams.size();
    if (nside) {
        PLAN_HIP(hipEventRecord(p->entry, s0));   // side lanes start behind whatever the caller's stream already holds
        for (hipStream_t s : p->streams) PLAN_HIP(hipStreamWaitEvent(s, p->entry, 0));
    }
    for (const Toc3dPlan::Node& n : p->nodes) {
        hipStream_t s = n.lane == 0 ? s0 : p->streams[n.lane - 1];
        for (int w : n.waits) PLAN_HIP(hipStreamWaitEvent(s, p->events[w], 0));
        PLAN_HIP(hipLaunchKernel(n.func, n.grid, n.block, p->argv.data() + n.first_arg, n.lds, s));
        if (n.signal >= 0) PLAN_HIP(hipEventRecord(p->events[n.signal], s));
    }
    for (size_t l = 0; l < nside; ++l) {          // join: the caller's stream ends behind every lane
        PLAN_HIP(hipEventRecord(p->lane_done[l], p->streams[l]));
        PLAN_HIP(hipStreamWaitEvent(s0, p->lane_done[l], 0));
    }
    return TOC3D_OK;
}

}  // extern "C"
