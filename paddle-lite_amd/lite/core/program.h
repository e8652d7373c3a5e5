// program.h — Instruction / RuntimeProgram (lite/core/program.{h,cc}): an ordered list of {OpLite, KernelBase};
// Instruction::Run = CheckShape once; op->InferShape(); kernel->Launch()  (program.cc:436-467);
// RuntimeProgram::Run runs them sequentially on the caller's thread (program.cc:265-315).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "lite/core/op_lite.h"

namespace paddle {
namespace lite {

struct Instruction {
  Instruction(std::shared_ptr<OpLite> op, std::unique_ptr<KernelBase>&& kernel)
      : op_(std::move(op)), kernel_(std::move(kernel)) {}
  void Run() {
    if (first_epoch_) {
      first_epoch_ = false;
      CHECK(op_->CheckShape());
    }
    op_->InferShape();
    kernel_->Launch();
  }
  OpLite* op() { return op_.get(); }
  KernelBase* kernel() { return kernel_.get(); }
  bool is_io_copy() const { return op_->Type() == "io_copy"; }

 private:
  std::shared_ptr<OpLite> op_;
  std::unique_ptr<KernelBase> kernel_;
  bool first_epoch_{true};
};

class RuntimeProgram {
 public:
  void Add(Instruction&& inst) { insts_.emplace_back(std::move(inst)); }
  // skip_io_copy: run with the feed already resident on the device (bench: "inputs resident in HBM").
  void Run(bool skip_io_copy = false) {
    for (auto& i : insts_) {
      if (skip_io_copy && i.is_io_copy()) continue;
      i.Run();
    }
  }
  std::vector<Instruction>& instructions() { return insts_; }

 private:
  std::vector<Instruction> insts_;
};

}  // namespace lite
}  // namespace paddle
