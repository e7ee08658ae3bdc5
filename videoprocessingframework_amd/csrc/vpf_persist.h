// vpf_persist.h — host side of the persistent launches (k_resize_common.h: k_planes_mp_persist; k_lanczos_mfma.hip): which eight work
// counters a launch uses, how many workgroups stay resident, and how the item list is cut into the XCDs' shares.  No HIP in the slot
// table itself (tests/test_persist_cpu.py compiles it with g++); the launchers pass the two stream questions in as callbacks.
//
// Counters: a block of eight per SLOT in static device memory, zero when the code object is loaded and zero again after every launch (the
// wave that draws a counter's last ticket puts it back) — nothing for the host to track, nothing a device reset can leave stale.  Two
// launches may share a slot only if they cannot run at the same time: a slot belongs to ONE stream (launches of a stream run in order).
// A stream meets the table once and keeps its slot; when the table is full, a slot whose stream is idle (or gone) is handed on; when none
// is, the launch simply is not persistent.  Captured launches are never persistent (a graph replays on any stream at any time).
#pragma once
#include <stdint.h>

#include <mutex>

namespace vpf {

constexpr int kPersistSlots = 64;
class PersistSlotTable {
 public:
  // stream_idle(stream) -> true when nothing is queued or running on it any more (or the handle no longer names a stream)
  template <class Idle>
  int slot_of(int dev, const void* stream, Idle&& stream_idle) {
    std::lock_guard<std::mutex> g(mu_);
    tick_++;
    int free_slot = -1, oldest = -1;
    for (int i = 0; i < kPersistSlots; i++) {
      Ent& e = e_[i];
      if (e.used && e.dev == dev && e.stream == stream) { e.tick = tick_; return i; }
      if (!e.used) { if (free_slot < 0) free_slot = i; }
      else if (oldest < 0 || e.tick < e_[oldest].tick) oldest = i;
    }
    if (free_slot < 0) {  // full: the least recently used slot, if its stream has drained (its counters are back at zero then)
      if (oldest < 0 || !stream_idle(e_[oldest].dev, e_[oldest].stream)) return -1;
      free_slot = oldest;
    }
    e_[free_slot] = Ent{true, dev, stream, tick_};
    return free_slot;
  }
  int used() {
    std::lock_guard<std::mutex> g(mu_);
    int n = 0;
    for (const Ent& e : e_) n += e.used;
    return n;
  }

 private:
  struct Ent { bool used; int dev; const void* stream; uint64_t tick; };
  std::mutex mu_;
  Ent e_[kPersistSlots] = {};
  uint64_t tick_ = 0;
};

// the XCDs' shares of `total` picture-ordered items: eight contiguous ranges, sizes differing by at most one
inline void persist_shares(uint32_t total, uint32_t lo[9]) {
  const uint32_t per = total / 8u, rem = total % 8u;
  lo[0] = 0;
  for (uint32_t x = 0; x < 8; x++) lo[x + 1] = lo[x] + per + (x < rem ? 1u : 0u);
}

}  // namespace vpf
