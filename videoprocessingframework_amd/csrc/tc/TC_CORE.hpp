// TC_CORE.hpp — Token / Task: the unit-of-work base of the Task layer.
// Same public surface as the reference's TC_CORE (src/TC/TC_CORE/inc/TC_CORE.hpp:26-109): a Task owns
// fixed-size vectors of non-owning Token* inputs/outputs, Run() is the overridable body and Execute()
// is Run() followed by the optional synchronisation callback (src/TC/TC_CORE/src/Task.cpp:50-57).
// tests/test_tc_core_vs_reference.py drives this class and the reference's own TC_CORE (compiled from
// /root/reference by oracle/Makefile into oracle/_ref) through the same call sequences.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace VPF {

class Token {
public:
  Token(const Token&) = delete;
  Token& operator=(const Token&) = delete;
  virtual ~Token() = default;

protected:
  Token() = default;
};

enum class TaskExecStatus { TASK_EXEC_SUCCESS, TASK_EXEC_FAIL };

// called after Run() by a blocking task (e.g. a stream synchronise)
typedef void (*p_sync_call)(void* p_args);

class Task {
public:
  Task() = delete;
  Task(const Task&) = delete;
  Task& operator=(const Task&) = delete;
  virtual ~Task() = default;

  virtual TaskExecStatus Run() { return TaskExecStatus::TASK_EXEC_SUCCESS; }
  virtual TaskExecStatus Execute() {
    const TaskExecStatus ret = Run();
    if (sync_ && sync_args_) sync_(sync_args_);
    return ret;
  }

  bool SetInput(Token* t, uint32_t i) { return assign(inputs_, t, i); }
  bool SetOutput(Token* t, uint32_t i) { return assign(outputs_, t, i); }
  void ClearInputs() { inputs_.assign(inputs_.size(), nullptr); }
  void ClearOutputs() { outputs_.assign(outputs_.size(), nullptr); }
  Token* GetInput(uint32_t i = 0) { return i < inputs_.size() ? inputs_[i] : nullptr; }
  Token* GetOutput(uint32_t i = 0) { return i < outputs_.size() ? outputs_[i] : nullptr; }
  uint64_t GetNumInputs() const { return inputs_.size(); }
  uint64_t GetNumOutputs() const { return outputs_.size(); }
  const char* GetName() const { return name_.c_str(); }

protected:
  Task(const char* name, uint32_t num_inputs, uint32_t num_outputs, p_sync_call sync = nullptr, void* sync_args = nullptr)
      : name_(name), inputs_(num_inputs, nullptr), outputs_(num_outputs, nullptr), sync_(sync), sync_args_(sync_args) {}

private:
  static bool assign(std::vector<Token*>& v, Token* t, uint32_t i) {
    if (i >= v.size()) return false;
    v[i] = t;
    return true;
  }
  std::string name_;
  std::vector<Token*> inputs_, outputs_;
  p_sync_call sync_;
  void* sync_args_;
};

}  // namespace VPF
