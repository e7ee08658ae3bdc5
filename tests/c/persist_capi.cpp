// C wrapper around vpf_persist.h for tests/test_persist_cpu.py (g++, no HIP): the stream -> counter-slot table of the persistent launches and
// the XCDs' shares of an item list
#include "vpf_persist.h"

#include <set>

namespace {
struct Table {
  vpf::PersistSlotTable t;
  std::set<uint64_t> busy;  // streams the test declares "not drained"
};
}  // namespace

extern "C" {
void* pst_new() { return new Table(); }
void pst_free(void* p) { delete static_cast<Table*>(p); }
void pst_busy(void* p, uint64_t stream, int on) { Table* T = static_cast<Table*>(p); if (on) T->busy.insert(stream); else T->busy.erase(stream); }
int pst_slot(void* p, int dev, uint64_t stream) {
  Table* T = static_cast<Table*>(p);
  return T->t.slot_of(dev, reinterpret_cast<const void*>(stream), [T](int, const void* s) { return T->busy.count((uint64_t)(uintptr_t)s) == 0; });
}
int pst_take(void* p, int dev, uint64_t stream, int* set) {
  Table* T = static_cast<Table*>(p);
  return T->t.take(dev, reinterpret_cast<const void*>(stream), [T](int, const void* s) { return T->busy.count((uint64_t)(uintptr_t)s) == 0; }, set);
}
int pst_used(void* p) { return static_cast<Table*>(p)->t.used(); }
void pst_shares(uint32_t total, uint32_t* lo) { vpf::persist_shares(total, lo); }
}
