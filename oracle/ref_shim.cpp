// C shim over the REFERENCE's own TC_CORE (compiled from /root/reference/src/TC/TC_CORE by
// oracle/Makefile).  Lets tests drive the real VPF::Task / VPF::Token state machine and compare it
// with our re-implementation (videoprocessingframework_amd/csrc/tc_core) call for call.
// Test infrastructure only.
#include "TC_CORE.hpp"
#include <cstdint>
using namespace VPF;
namespace {
struct RefToken final : public Token {
  RefToken() = default;
};
struct RefTask final : public Task {
  RefTask(const char* n, uint32_t ni, uint32_t no, p_sync_call c, void* a, int ret)
      : Task(n, ni, no, c, a), ret_(ret) {}
  TaskExecStatus Run() override {
    runs++;
    return ret_ ? TaskExecStatus::TASK_EXEC_FAIL : TaskExecStatus::TASK_EXEC_SUCCESS;
  }
  int runs = 0;
  int ret_;
};
void bump(void* p) { ++*static_cast<int*>(p); }
}  // namespace
extern "C" {
void* ref_token_new() { return new RefToken; }
void ref_token_del(void* t) { delete static_cast<RefToken*>(t); }
// sync_counter may be null (=> no sync call registered, like ConvertSurface)
void* ref_task_new(const char* name, uint32_t ni, uint32_t no, int* sync_counter, int run_ret) {
  return new RefTask(name, ni, no, sync_counter ? bump : nullptr, sync_counter, run_ret);
}
void ref_task_del(void* t) { delete static_cast<RefTask*>(t); }
int ref_task_set_input(void* t, void* tok, uint32_t i) { return static_cast<RefTask*>(t)->SetInput(static_cast<Token*>(tok), i); }
int ref_task_set_output(void* t, void* tok, uint32_t i) { return static_cast<RefTask*>(t)->SetOutput(static_cast<Token*>(tok), i); }
void* ref_task_get_input(void* t, uint32_t i) { return static_cast<RefTask*>(t)->GetInput(i); }
void* ref_task_get_output(void* t, uint32_t i) { return static_cast<RefTask*>(t)->GetOutput(i); }
void ref_task_clear_inputs(void* t) { static_cast<RefTask*>(t)->ClearInputs(); }
void ref_task_clear_outputs(void* t) { static_cast<RefTask*>(t)->ClearOutputs(); }
uint64_t ref_task_num_inputs(void* t) { return static_cast<RefTask*>(t)->GetNumInputs(); }
uint64_t ref_task_num_outputs(void* t) { return static_cast<RefTask*>(t)->GetNumOutputs(); }
const char* ref_task_name(void* t) { return static_cast<RefTask*>(t)->GetName(); }
int ref_task_execute(void* t) { return static_cast<RefTask*>(t)->Execute() == TaskExecStatus::TASK_EXEC_SUCCESS ? 0 : 1; }
int ref_task_runs(void* t) { return static_cast<RefTask*>(t)->runs; }
}
