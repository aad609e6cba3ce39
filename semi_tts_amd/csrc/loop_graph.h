// loop_graph.h -- hipGraph replay of the decoder loops of a TRAINING step (st_decoder_forward / st_decoder_backward).
//
// A training step issues 255 + 255 launches from the two C++ loops at ~4.7 us of host time each.  Once no allocation outlives a step
// the caching allocator hands every tensor of a step the address it had in the step before, so the loops are
// called with bit-identical arguments step after step: the launches of such a call are captured into a hipGraph at the second sighting
// of an argument set (on the calling stream, thread-local capture) and replayed with ONE launch whenever the same arguments come back.
// Any other argument set -- another shape, another address -- runs eagerly as before.  Round 5 measured this with the step bound by
// the GPU (no gain: a replay's kernels start ~0.2 us later each); round 6 measured it again at small batches (the GPU bounds those steps too)
// and keeps it as an option: ST_LOOP_GRAPHS=1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>

namespace stlg {

struct Entry { uint64_t key; hipGraphExec_t exec; int seen; unsigned long long age; int dev; hipStream_t stream; };
constexpr int SLOTS = 12;
struct Cache {
    Entry e[SLOTS] = {};
    unsigned long long age = 0;
    long replays = 0, captures = 0, eager = 0;
    hipStream_t cap = nullptr;          // the capture runs on a stream of the cache's own: torch's default stream is the legacy stream, which cannot capture
    std::mutex mu;
};

inline uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* c = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}
constexpr uint64_t FNV0 = 1469598103934665603ull;

int& enabled_flag();                                   // (one flag for the library: defined in decoder_bwd.hip)
// 1 / 0: on / off (ST_LOOP_GRAPHS, st_loop_graphs_enable); OFF unless asked for.  Measured (round 6): the replay takes
// 0.8 ms of HOST time off a C2 training step (8.1 -> 7.3 ms) and 1.2 ms off a text-first cycle step at B = 8 (10.0 -> 8.8 ms) -- and nothing off
// either step: both are bound by the GPU's chains of dependent launches (8.78 -> 8.90 ms; 10.0 -> 10.1 ms), and a replay's kernels start a
// little later each.  Kept for hosts slower than the measured ones.
inline bool enabled(int B = 0) {
    (void)B;
    int& v = enabled_flag();
    if (v < 0) { const char* e = getenv("ST_LOOP_GRAPHS"); v = e && atoi(e) ? 1 : 0; }
    return v == 1;
}

// 1: replayed, nothing left to do.  2: a capture has begun on *issue_on -- the caller issues its launches THERE and calls end().  0: the caller
// issues its launches on `st` as usual (first sighting, an outer capture in progress, a HIP error, the cache switched off).
inline int begin(Cache& c, uint64_t key, hipStream_t st, Entry** out, hipStream_t* issue_on) {
    *out = nullptr;
    *issue_on = st;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 0; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lock(c.mu);
    Entry* hit = nullptr;
    Entry* victim = &c.e[0];
    for (int i = 0; i < SLOTS; ++i) {
        Entry& e = c.e[i];
        if (e.seen != 0 && e.key == key && e.dev == dev && e.stream == st) { hit = &e; break; }
        if (e.age < victim->age) victim = &e;
    }
    if (hit && hit->exec) {
        hit->age = ++c.age;
        if (hipGraphLaunch(hit->exec, st) == hipSuccess) { ++c.replays; return 1; }
        (void)hipGetLastError();
        (void)hipGraphExecDestroy(hit->exec);
        hit->exec = nullptr; hit->seen = -1;           // never again for this argument set
        ++c.eager;
        return 0;
    }
    if (hit && hit->seen > 0) {                        // second sighting: capture
        hit->age = ++c.age;
        if (!c.cap && hipStreamCreateWithFlags(&c.cap, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c.cap = nullptr; }
        if (!c.cap || hipStreamBeginCapture(c.cap, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); hit->seen = -1; ++c.eager; return 0; }
        *out = hit;
        *issue_on = c.cap;
        return 2;
    }
    if (!hit) {
        if (victim->exec) (void)hipGraphExecDestroy(victim->exec);
        *victim = Entry{key, nullptr, 1, ++c.age, dev, st};
    }
    ++c.eager;
    return 0;
}

// ends the capture begun by begin() == 2 and launches the graph; rc_issue = what the issuing code returned
inline int end(Cache& c, Entry* e, hipStream_t st, int rc_issue) {
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(c.cap, &g);
    std::lock_guard<std::mutex> lock(c.mu);
    if (rc_issue != 0 || ce != hipSuccess || !g) {
        (void)hipGetLastError();
        if (g) (void)hipGraphDestroy(g);
        e->seen = -1;
        return rc_issue != 0 ? rc_issue : -5;
    }
    hipGraphExec_t x = nullptr;
    if (hipGraphInstantiate(&x, g, nullptr, nullptr, 0) != hipSuccess || !x) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(g);
        e->seen = -1;
        return -5;       // (the launches were captured, not run: the caller reports the failure)
    }
    (void)hipGraphDestroy(g);
    e->exec = x;
    ++c.captures;
    if (hipGraphLaunch(x, st) != hipSuccess) { (void)hipGetLastError(); return -5; }
    return 0;
}

}  // namespace stlg
