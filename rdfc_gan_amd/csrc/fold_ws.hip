// fold_ws.hip -- arrival counters for the in-launch second stage of the two-stage reductions (fold.h).
//
// A launch that folds its finalize needs a handful of agent-scope counters (<= ~100) that are ZERO when its first workgroup
// arrives.  The last arriver of every counter zeroes it again (fold.h), so a slot never needs a memset -- it only must not be handed
// to two launches that can be in flight at the same time.  One allocation per device, zeroed once:
//   * eager launches take consecutive slots from a ring of RING counters (4 MB): a slot comes round again after >= 10 000 launches;
//     the host is never more than a step (~400 such launches) ahead of the device because every step reads its losses back;
//   * launches recorded into a HIP graph (hipStreamIsCapturing) take slots from a second region that is never recycled -- a replay
//     re-uses its own slots (replays of one graph are ordered), and nothing eager can ever alias them.
#include "fold.h"

#include <atomic>
#include <mutex>

namespace {
constexpr size_t RING = 1u << 20, CAPTURE = 1u << 20;
struct Ws {
    unsigned *base = nullptr;
    std::atomic<size_t> head{0}, cap_head{0};
};
Ws g_ws[64];
std::mutex g_mu;
}  // namespace

unsigned *rdfc_fold_counters(int n, hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        rdfc_set_error("rdfc_fold_counters: no current device");
        return nullptr;
    }
    Ws &w = g_ws[dev];
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
    if (w.base == nullptr) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (w.base == nullptr) {
            if (capturing) {
                rdfc_set_error("rdfc_fold_counters: first use inside a stream capture (run one eager step before capturing)");
                return nullptr;
            }
            void *p = nullptr;
            if (hipMalloc(&p, (RING + CAPTURE) * sizeof(unsigned)) != hipSuccess || hipMemset(p, 0, (RING + CAPTURE) * sizeof(unsigned)) != hipSuccess) {
                rdfc_set_error("rdfc_fold_counters: cannot allocate the counter ring");
                return nullptr;
            }
            w.base = (unsigned *)p;
        }
    }
    const size_t need = ((size_t)n + 31) & ~(size_t)31;          // whole 128-byte lines per launch
    if (need > RING / 4) {
        rdfc_set_error("rdfc_fold_counters: %d counters in one launch", n);
        return nullptr;
    }
    if (capturing) {
        const size_t at = w.cap_head.fetch_add(need);
        if (at + need > CAPTURE) {
            rdfc_set_error("rdfc_fold_counters: capture region exhausted (%zu counters recorded into graphs)", at);
            return nullptr;
        }
        return w.base + RING + at;
    }
    for (;;) {
        size_t at = w.head.load();
        size_t start = at % RING;
        if (start + need > RING) start = 0;                        // do not straddle the end of the ring
        const size_t next = (at - at % RING) + (start == 0 && at % RING != 0 ? RING : 0) + start + need;
        if (w.head.compare_exchange_weak(at, next)) return w.base + start;
    }
}
