// vpf_persist.h — host side of the persistent launches (k_resize_common.h: k_planes_mp_persist; k_lanczos_mfma.hip): which eight work
// counters a launch uses, how many workgroups stay resident, and how the item list is cut into the XCDs' shares.  No HIP in the slot
// table itself (tests/test_persist_cpu.py compiles it with g++); the launchers pass the two stream questions in as callbacks.
//
// Counters: TWO sets of eight per SLOT in static device memory, all zero when the code object is loaded.  A launch draws from one set and puts
// the OTHER back to zero; the slot's next launch draws from that other set (take() hands out the sets in turn).  Two launches may share a
// slot only if they cannot run at the same time: a slot belongs to ONE stream (launches of a stream run in order), so the set a launch
// zeroes is the one the launch before it has finished with.  (Round 5's single set was put back by "the wave that draws the last ticket",
// which needed every wave to fail exactly once on every counter: 8 x 4 096 atomics at the end of each launch.)
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
  // the slot of (dev, stream) and, in *set, which of its two counter sets this launch draws from (the other one is the one it zeroes)
  template <class Idle>
  int take(int dev, const void* stream, Idle&& stream_idle, int* set) {
    const int s = slot_of(dev, stream, stream_idle);
    if (s >= 0) {
      std::lock_guard<std::mutex> g(mu_);
      *set = e_[s].set;
      e_[s].set ^= 1;
    }
    return s;
  }
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
    e_[free_slot] = Ent{true, dev, stream, tick_, e_[free_slot].set};  // (the sets keep their turn when a slot changes hands: its last launch zeroed the one that is next)
    return free_slot;
  }
  int used() {
    std::lock_guard<std::mutex> g(mu_);
    int n = 0;
    for (const Ent& e : e_) n += e.used;
    return n;
  }

 private:
  struct Ent { bool used; int dev; const void* stream; uint64_t tick; int set; };
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
