// tsan_harness.cpp — the host-side concurrency of the launch bookkeeping under ThreadSanitizer (tools/sanitize.sh): eight
// threads, each with a stream of its own (and one stream they all share), cycle plane shapes through an LzmTableCache whose arena holds four
// tables — lookups, builds, evictions behind events, "used" marks — and ask the PersistSlotTable for counter slots from more streams than it has
// slots.  The HIP events are a thread-safe stand-in (the cache calls it under its own lock; the stand-in is also touched from outside it).
// Exit status 0 and no ThreadSanitizer report = pass.  g++ -std=c++17 -O1 -g -fsanitize=thread -pthread
#include "vpf_lzm_plan.h"
#include "vpf_persist.h"

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

namespace {
struct CountingSync final : vpf::LzmSync {
  std::atomic<uint64_t> next{1}, completed{0};
  std::atomic<int> live{0};
  void* record(const void*, int) override { live++; return reinterpret_cast<void*>((uintptr_t)next++); }
  bool done(void* ev) override { return (uint64_t)(uintptr_t)ev <= completed.load(); }
  void wait(const void*, void*) override {}
  void destroy(void*) override { live--; }
  bool device_alive(int) override { return true; }
  int stream_state(const void* s) override { return ((uintptr_t)s & 0xf00) == 0xf00 ? 1 : 0; }  // streams 0x..f.. "are capturing"
};
}  // namespace

int main() {
  CountingSync sync;
  vpf::LzmTableCache cache(256 + 4 * 4096, &sync);
  vpf::PersistSlotTable slots;
  std::atomic<uint64_t> hits{0}, builds{0}, none{0}, slot_ok{0}, slot_none{0};
  std::atomic<bool> stop{false};
  std::vector<std::thread> ts;
  for (int t = 0; t < 8; t++)
    ts.emplace_back([&, t] {
      const void* mine = reinterpret_cast<const void*>((uintptr_t)(0x1000 + t));
      const void* shared = reinterpret_cast<const void*>((uintptr_t)0x9000);
      for (int i = 0; i < 20000; i++) {
        const void* st = (i & 7) == 0 ? shared : mine;
        const uint32_t shape = (uint32_t)((i * 7 + t * 13) % 11);  // eleven shapes through four places
        cache.begin(0);
        int ids[2];
        const vpf::LzmTableCache::Hit a = cache.get(st, 0, false, 0, shape, 1, 1, 8, 4096);
        if (a.off16 && a.build) cache.built(a.id, st, false);
        const vpf::LzmTableCache::Hit b = cache.get(st, 0, false, 1, shape, 1, 1, 0, 4096);
        if (b.off16 && b.build) cache.built(b.id, st, false);
        ids[0] = a.id; ids[1] = b.id;
        cache.used(st, 0, ids, 2);
        cache.end();
        (a.off16 ? (a.build ? builds : hits) : none)++;
        if ((i & 63) == 0) sync.completed.store(sync.next.load() - 1);  // "the GPU has caught up"
        int set = -1;  // (round 6: a slot hands out its two counter sets in turn — take() — under the same lock)
        const int s = slots.take(0, reinterpret_cast<const void*>((uintptr_t)(0x20000 + (i * 5 + t) % 70)), [&](int, const void* q) { return ((uintptr_t)q & 3) != 0; }, &set);
        if (s >= 0 && set != 0 && set != 1) return;
        (s >= 0 ? slot_ok : slot_none)++;
      }
    });
  for (auto& th : ts) th.join();
  cache.begin(-1);
  const uint32_t entries = cache.entries(0);
  cache.end();
  std::printf("tsan harness: table lookups %llu hits, %llu builds, %llu without a table; %u entries live, %d events live; counter slots %llu given, %llu refused, %d in use\n",
              (unsigned long long)hits.load(), (unsigned long long)builds.load(), (unsigned long long)none.load(), entries, sync.live.load(),
              (unsigned long long)slot_ok.load(), (unsigned long long)slot_none.load(), slots.used());
  return entries <= 4 && slots.used() <= vpf::kPersistSlots ? 0 : 1;
}
