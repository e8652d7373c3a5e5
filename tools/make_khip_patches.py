#!/usr/bin/env python3
"""Authoring tool of patches/*.patch: the kHIP touch-points inside a real Paddle-Lite tree (SURVEY.md 8b / 8f rank 3).
For every touched file of the reference (read as text from --reference, never copied into this repo) it applies a list
of exact-string edits to a scratch copy under /tmp and writes the unified diff.  tools/check_patches.sh proves that the
committed patches still apply (`git apply --check`) to the reference tree.

Usage: python tools/make_khip_patches.py [--reference /root/reference]"""
import argparse
import difflib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HIP_MEMCPY = '''  } else if (type == TargetType::kHIP) {
#ifdef LITE_WITH_HIP
    lite::TargetWrapperHip::MemcpySync(
        data, src_data, num * sizeof(T), lite::IoDirection::%s);
#else
    LOG(FATAL) << "Please compile the lib with HIP.";
#endif
'''

PASS_H = '// hip_conv_tail_fuse_pass.h — kHIP graph-level conv-tail fusions as a mir pass (see hip_conv_tail_matcher.h).\n#pragma once\n\n#include <memory>\n#include "lite/core/mir/pass.h"\n\nnamespace paddle {\nnamespace lite {\nnamespace mir {\n\nclass HipConvTailFusePass : public ProgramPass {\n public:\n  void Apply(const std::unique_ptr<SSAGraph>& graph) override;\n};\n\n}  // namespace mir\n}  // namespace lite\n}  // namespace paddle\n'

PASS_CC = '// hip_conv_tail_fuse_pass.cc — SSAGraph adapter of the kHIP conv-tail matcher.\n//\n// Runs behind type_precision_cast_pass / type_layout_cast_pass (the calib statements exist, every statement has its picked\n// kernel) and in front of runtime_context_assign_pass.  It lists the statements in topological order for\n// fusion::MatchConvTails, then applies its decisions the way lite/core/mir/fusion/conv_elementwise_fuser.cc applies its\n// own: the surviving conv statement gets the residual operand through the reference\'s ConvParam::residualData\n// (op input "ResidualData" + attribute fuse_residual_connection, lite/operators/conv_op.h:102), its output variable is\n// re-linked, the fused statements are removed with GraphSafeRemoveNodes, and the picked kHIP kernel receives the state the\n// reference has no field for (relu behind the add, the int8 calib copy, the dropped fp32 output) through\n// kernels::hip::HipFusableKernel::SetFusion (lite/kernels/hip/conv_fusion.h).\n#include "lite/core/mir/fusion/hip_conv_tail_fuse_pass.h"\n#include <map>\n#include <set>\n#include <string>\n#include <vector>\n#include "lite/core/mir/fusion/hip_conv_tail_matcher.h"\n#include "lite/core/mir/pass_registry.h"\n#include "lite/core/mir/pattern_matcher.h"\n#include "lite/kernels/hip/conv_fusion.h"\n\nnamespace paddle {\nnamespace lite {\nnamespace mir {\n\nnamespace {\n\nstd::string first_arg(const OpInfo* info, const std::string& slot, bool input) {\n  const auto& names = input ? info->Input(slot) : info->Output(slot);\n  return names.empty() ? std::string() : names.front();\n}\n\n// one statement as the matcher sees it\nfusion::TailInst Describe(Node* n) {\n  fusion::TailInst t;\n  auto& stmt = n->AsStmt();\n  const OpInfo* info = stmt.op_info();\n  const std::string type = stmt.op_type();\n  auto& kernel = stmt.picked_kernel();\n  const bool hip = kernel.target() == TARGET(kHIP);\n  if (type == "conv2d" && hip && kernel.precision() == PRECISION(kInt8) && kernel.alias() == "fp32_out") {\n    t.kind = fusion::TailInst::kConvF32;\n    t.inputs = {first_arg(info, "Input", true)};\n    t.output = first_arg(info, "Output", false);\n  } else if ((type == "elementwise_add" || type == "fusion_elementwise_add_activation") && hip) {\n    const bool relu = type == "fusion_elementwise_add_activation" && info->GetAttr<std::string>("act_type") == "relu";\n    t.kind = type == "elementwise_add" ? fusion::TailInst::kAdd : (relu ? fusion::TailInst::kAddRelu : fusion::TailInst::kOther);\n    t.inputs = {first_arg(info, "X", true), first_arg(info, "Y", true)};\n    t.output = first_arg(info, "Out", false);\n  } else if (type == "pool2d" && hip && info->GetAttr<std::string>("pooling_type") == "max") {\n    t.kind = fusion::TailInst::kMaxPool;\n    t.inputs = {first_arg(info, "X", true)};\n    t.output = first_arg(info, "Out", false);\n  } else if (type == "calib" && hip && kernel.alias() == "fp32_to_int8") {\n    t.kind = fusion::TailInst::kCalibF2I;\n    t.inputs = {first_arg(info, "Input", true)};\n    t.output = first_arg(info, "Out", false);\n    t.calib_scale = info->GetAttr<float>("scale");\n  } else {  // every other statement only counts as a reader / writer of its variables\n    for (auto* in : n->inlinks) t.inputs.push_back(in->AsArg().name);\n    t.output = n->outlinks.empty() ? std::string() : n->outlinks.front()->AsArg().name;\n  }\n  return t;\n}\n\n// ResetOp creates the statement\'s kernels again for every valid place: keep the one this pass wants (static_kernel_pick_pass has\n// run already and will not run again)\nvoid Repick(Node::Stmt* stmt, PrecisionType precision, const std::string& alias) {\n  std::vector<std::unique_ptr<KernelBase>> keep;\n  for (auto& k : stmt->kernels()) {\n    if (k->target() == TARGET(kHIP) && k->precision() == precision && k->alias() == alias) {\n      keep.emplace_back(std::move(k));\n      break;\n    }\n  }\n  CHECK(!keep.empty()) << stmt->op_type() << ": no kHIP kernel " << alias;\n  stmt->SetKernels(std::move(keep));\n}\n\n}  // namespace\n\nvoid HipConvTailFusePass::Apply(const std::unique_ptr<SSAGraph>& graph) {\n  std::vector<Node*> order = graph->StmtTopologicalOrder();\n  std::vector<fusion::TailInst> before, prog;\n  for (auto* n : order) before.push_back(Describe(n));\n  prog = before;\n  fusion::MatchConvTails(&prog);\n\n  std::set<const Node*> dead;\n  for (size_t i = 0; i < order.size(); ++i) {\n    Node* n = order[i];\n    const fusion::TailInst& t = prog[i];\n    if (t.dead) {\n      dead.insert(n);\n      continue;\n    }\n    auto& stmt = n->AsStmt();\n    if (t.kind == fusion::TailInst::kMaxPool && t.pool_int8) {\n      // pool2d(max) moved behind the quantiser: reads the conv\'s int8 copy, writes the former calib output; the int8 kernel is\n      // picked again from the valid places (pool2d kHIP kInt8 def)\n      cpp::OpDesc desc = *stmt.mutable_op_info();\n      desc.SetInput("X", {t.inputs[0]});\n      desc.SetOutput("Out", {t.output});\n      stmt.ResetOp(desc, graph->valid_places());\n      Repick(&stmt, PRECISION(kInt8), "def");\n      for (auto* in : std::vector<Node*>(n->inlinks.begin(), n->inlinks.end())) RemoveDirectedLink(in, n);\n      for (auto* out : std::vector<Node*>(n->outlinks.begin(), n->outlinks.end())) RemoveDirectedLink(n, out);\n      IR_NODE_LINK_TO(graph->RetrieveArgument(t.inputs[0]) ? graph->RetrieveArgument(t.inputs[0]) : graph->NewArgumentNode(t.inputs[0]), n);\n      IR_OP_VAR_LINK(n, graph->RetrieveArgument(t.output));\n      continue;\n    }\n    if (t.kind != fusion::TailInst::kConvF32 || (t.residual.empty() && t.calib_out.empty())) continue;\n    // ---- the surviving conv\n    cpp::OpDesc desc = *stmt.mutable_op_info();\n    auto* scope = stmt.op()->scope();\n    if (!t.residual.empty()) {\n      desc.SetInput("ResidualData", {t.residual});\n      desc.SetAttr("fuse_residual_connection", true);\n      IR_NODE_LINK_TO(graph->RetrieveArgument(t.residual), n);\n    }\n    if (t.output != before[i].output) {  // the conv writes the sum now\n      desc.SetOutput("Output", {t.output});\n      for (auto* out : std::vector<Node*>(n->outlinks.begin(), n->outlinks.end())) {\n        RemoveDirectedLink(n, out);\n        dead.insert(out);  // the conv\'s old output variable has no reader left\n      }\n      IR_OP_VAR_LINK(n, graph->RetrieveArgument(t.output));\n    }\n    stmt.ResetOp(desc, graph->valid_places());  // re-attaches the param (residualData)\n    Repick(&stmt, PRECISION(kInt8), "fp32_out");\n    kernels::hip::HipConvFusion f;\n    f.fuse_residual_relu = t.residual_relu;\n    if (!t.calib_out.empty()) {\n      Node* q = graph->RetrieveArgument(t.calib_out);\n      if (!q) q = graph->NewArgumentNode(t.calib_out);  // pattern (C): "<conv out>/precision_trans" is new\n      IR_OP_VAR_LINK(n, q);\n      f.calib_output = scope->Var(t.calib_out)->GetMutable<lite::Tensor>();\n      f.calib_scale = t.fused_calib_scale;\n      f.drop_fp32_output = t.drop_f32;\n    }\n    auto* fusable = dynamic_cast<kernels::hip::HipFusableKernel*>(&stmt.picked_kernel());\n    CHECK(fusable) << "conv2d kHIP kInt8 fp32_out must implement HipFusableKernel";\n    fusable->SetFusion(f);\n  }\n  GraphSafeRemoveNodes(graph.get(), dead);\n}\n\n}  // namespace mir\n}  // namespace lite\n}  // namespace paddle\n\nREGISTER_MIR_PASS(hip_conv_tail_fuse_pass, paddle::lite::mir::HipConvTailFusePass).BindTargets({TARGET(kHIP)});\n'

EDITS = {
    "0001-place-add-kHIP-target.patch": {
        "lite/api/paddle_place.h": [
            ("  kImaginationNNA = 15,\n  NUM = 16,  // number of fields.\n",
             "  kImaginationNNA = 15,\n  kHIP = 16,  // AMD Instinct (gfx950) through HIP: lite/backends/hip, lite/kernels/hip\n"
             "  NUM = 17,  // number of fields.\n"),
        ],
        "lite/api/paddle_place.cc": [
            ('                                              "huawei_ascend_npu",\n                                              "imagination_nna"};',
             '                                              "huawei_ascend_npu",\n                                              "imagination_nna",\n'
             '                                              "hip"};'),
            ('                                              "kHuaweiAscendNPU",\n                                              "kImaginationNNA"};',
             '                                              "kHuaweiAscendNPU",\n                                              "kImaginationNNA",\n'
             '                                              "kHIP"};'),
            ("                                               TARGET(kImaginationNNA)});",
             "                                               TARGET(kImaginationNNA),\n                                               TARGET(kHIP)});"),
        ],
    },
    "0002-memory-context-tensor-copy-for-kHIP.patch": {
        "lite/core/memory.cc": [
            ("#ifdef LITE_WITH_OPENCL\n    case TargetType::kOpenCL:\n      data = TargetWrapperCL::Malloc(size);",
             "#ifdef LITE_WITH_HIP\n    case TargetType::kHIP:\n      data = TargetWrapper<TARGET(kHIP)>::Malloc(size);\n      break;\n#endif  // LITE_WITH_HIP\n"
             "#ifdef LITE_WITH_OPENCL\n    case TargetType::kOpenCL:\n      data = TargetWrapperCL::Malloc(size);"),
            ("#ifdef LITE_WITH_OPENCL\n    case TargetType::kOpenCL:\n      if (free_flag == \"cl_use_image2d_\") {",
             "#ifdef LITE_WITH_HIP\n    case TargetType::kHIP:\n      TargetWrapper<TARGET(kHIP)>::Free(data);\n      break;\n#endif  // LITE_WITH_HIP\n"
             "#ifdef LITE_WITH_OPENCL\n    case TargetType::kOpenCL:\n      if (free_flag == \"cl_use_image2d_\") {"),
            ("#ifdef LITE_WITH_FPGA\n    case TargetType::kFPGA:\n      TargetWrapper<TARGET(kFPGA)>::MemcpySync(\n          dst, src, size, IoDirection::DtoD);\n      break;\n#endif",
             "#ifdef LITE_WITH_HIP\n    case TargetType::kHIP:\n      TargetWrapper<TARGET(kHIP)>::MemcpySync(\n          dst, src, size, IoDirection::DtoD);\n      break;\n#endif\n"
             "#ifdef LITE_WITH_FPGA\n    case TargetType::kFPGA:\n      TargetWrapper<TARGET(kFPGA)>::MemcpySync(\n          dst, src, size, IoDirection::DtoD);\n      break;\n#endif"),
        ],
        "lite/core/context.h": [
            ("#ifdef LITE_WITH_ARM\n      case TARGET(kARM):\n        kernel_contexts_[TargetType::kARM].As<ARMContext>().CopySharedTo(\n            &ctx->As<ARMContext>());\n        break;\n#endif",
             "#ifdef LITE_WITH_HIP\n      case TARGET(kHIP): {\n        // like kCUDA: one context per instruction, bound to the current device and its execution stream\n"
             "        int dev_id = TargetWrapper<TargetType::kHIP>::GetCurDevice();\n        ctx->As<HIPContext>().Init(dev_id, exec_stream_id);\n      } break;\n#endif\n"
             "#ifdef LITE_WITH_ARM\n      case TARGET(kARM):\n        kernel_contexts_[TargetType::kARM].As<ARMContext>().CopySharedTo(\n            &ctx->As<ARMContext>());\n        break;\n#endif"),
        ],
        "lite/api/paddle_api.cc": [
            ("        data, src_data, num * sizeof(T), lite::IoDirection::HtoD);\n#else\n    LOG(FATAL) << \"Please compile the lib with MLU.\";\n#endif\n  } else {\n    LOG(FATAL) << \"The CopyFromCpu interface just support kHost, kARM, kCUDA\";",
             "        data, src_data, num * sizeof(T), lite::IoDirection::HtoD);\n#else\n    LOG(FATAL) << \"Please compile the lib with MLU.\";\n#endif\n" + HIP_MEMCPY % "HtoD" +
             "  } else {\n    LOG(FATAL) << \"The CopyFromCpu interface just support kHost, kARM, kCUDA\";"),
            ("        data, src_data, num * sizeof(T), lite::IoDirection::DtoH);\n#else\n    LOG(FATAL) << \"Please compile the lib with MLU.\";\n#endif\n  } else {\n    LOG(FATAL) << \"The CopyToCpu interface just support kHost, kARM, kCUDA\";",
             "        data, src_data, num * sizeof(T), lite::IoDirection::DtoH);\n#else\n    LOG(FATAL) << \"Please compile the lib with MLU.\";\n#endif\n" + HIP_MEMCPY % "DtoH" +
             "  } else {\n    LOG(FATAL) << \"The CopyToCpu interface just support kHost, kARM, kCUDA\";"),
            ("template void Tensor::CopyFromCpu<int, TargetType::kMLU>(const int *);",
             "template void Tensor::CopyFromCpu<int, TargetType::kHIP>(const int *);\ntemplate void Tensor::CopyFromCpu<int64_t, TargetType::kHIP>(const int64_t *);\n"
             "template void Tensor::CopyFromCpu<float, TargetType::kHIP>(const float *);\ntemplate void Tensor::CopyFromCpu<uint8_t, TargetType::kHIP>(const uint8_t *);\n"
             "template void Tensor::CopyFromCpu<int8_t, TargetType::kHIP>(const int8_t *);\n\n"
             "template void Tensor::CopyFromCpu<int, TargetType::kMLU>(const int *);"),
        ],
    },
    "0003-optimizer-int8-place-and-activation-fusion-for-kHIP.patch": {
        "lite/api/cxx_api.cc": [
            ("  if (is_quantized_model) {\n    inner_places.insert(inner_places.begin(),\n                        Place{TARGET(kARM), PRECISION(kInt8)});\n  }",
             "  if (is_quantized_model) {\n    // the int8 place goes in front for the target the user asked for: kHIP when it is among the valid places,\n"
             "    // kARM otherwise (it was hard-coded)\n    bool has_hip = false;\n    for (auto &p : inner_places) has_hip = has_hip || p.target == TARGET(kHIP);\n"
             "    inner_places.insert(inner_places.begin(),\n                        Place{has_hip ? TARGET(kHIP) : TARGET(kARM), PRECISION(kInt8)});\n  }"),
        ],
        "lite/core/mir/fusion/conv_activation_fuse_pass.cc": [
            ("    if (place.target == TARGET(kARM)) {\n      has_arm = true;\n    }",
             "    // the kHIP int8 / fp32 conv epilogues implement relu6 and leaky_relu like the ARM ones do\n"
             "    if (place.target == TARGET(kARM) || place.target == TARGET(kHIP)) {\n      has_arm = true;\n    }"),
        ],
    },
    "0004-profile-DeviceTimer-kHIP-and-env-init.patch": {
        "lite/core/profile/timer.h": [
            ("}  // namespace profile\n}  // namespace lite\n}  // namespace paddle",
             "#ifdef LITE_WITH_HIP\n// hipEvents on the context's execution stream (lite/backends/hip/hip_timer.h holds the body: plhip_event_* of\n"
             "// include/plhip.h, so that this header needs no HIP headers)\ntemplate <>\nclass DeviceTimer<TargetType::kHIP> final : public Timer {\n public:\n"
             "  DeviceTimer();\n  ~DeviceTimer();\n  void Start(KernelContext* ctx);\n  float Stop(KernelContext* ctx);\n\n private:\n"
             "  void* owner_{nullptr};\n  void* e_start_{nullptr};\n  void* e_stop_{nullptr};\n};\n#endif\n\n"
             "}  // namespace profile\n}  // namespace lite\n}  // namespace paddle"),
        ],
        "lite/api/cxx_api_impl.cc": [
            ("#ifdef LITE_WITH_MLU\n    Env<TARGET(kMLU)>::Init();",
             "#ifdef LITE_WITH_HIP\n    for (auto &p : places) {\n      if (p.target == TARGET(kHIP)) {\n        Env<TARGET(kHIP)>::Init();  // enumerate the gfx950 devices (lite/core/device_info.h)\n"
             "        break;\n      }\n    }\n#endif\n#ifdef LITE_WITH_MLU\n    Env<TARGET(kMLU)>::Init();"),
        ],
    },
    "0005-runtime-sync-hooks-copysync-kernel-list-and-cmake-for-kHIP.patch": {
        "lite/core/program.h": [
            ("  void Sync() const { kernel_->mutable_context()->As<CUDAContext>().Sync(); }\n#endif\n",
             "  void Sync() const { kernel_->mutable_context()->As<CUDAContext>().Sync(); }\n#endif\n\n"
             "#ifdef LITE_WITH_HIP\n  // kHIP kernels enqueue on the context's execution stream; like kCUDA an instruction whose inputs were produced on\n"
             "  // another stream waits for them first (HIPContext::need_sync / Sync: lite/core/context.h)\n"
             "  bool need_sync_hip() const {\n    return kernel_->target() == TargetType::kHIP && kernel_->mutable_context()->As<HIPContext>().need_sync();\n  }\n"
             "  void SyncHip() const { kernel_->mutable_context()->As<HIPContext>().Sync(); }\n#endif\n"),
        ],
        "lite/core/program.cc": [
            ("#ifdef LITE_WITH_CUDA\n    if (inst.need_sync()) {\n      inst.Sync();\n    }\n#endif\n    inst.Run();",
             "#ifdef LITE_WITH_CUDA\n    if (inst.need_sync()) {\n      inst.Sync();\n    }\n#endif\n#ifdef LITE_WITH_HIP\n    if (inst.need_sync_hip()) {\n      inst.SyncHip();\n    }\n#endif\n    inst.Run();"),
        ],
        "lite/core/memory.h": [
            ("#ifdef LITE_WITH_OPENCL\n    case TargetType::kOpenCL:\n      TargetWrapperCL::MemcpySync(dst, src, size, dir);\n      break;\n#endif  // LITE_WITH_OPENCL\n",
             "#ifdef LITE_WITH_HIP\n    case TARGET(kHIP):\n      TargetWrapper<TARGET(kHIP)>::MemcpySync(dst, src, size, dir);\n      break;\n#endif\n"
             "#ifdef LITE_WITH_OPENCL\n    case TargetType::kOpenCL:\n      TargetWrapperCL::MemcpySync(dst, src, size, dir);\n      break;\n#endif  // LITE_WITH_OPENCL\n"),
        ],
        "lite/api/paddle_use_kernels.h": [
            ("USE_LITE_KERNEL(depthwise_conv2d, kARM, kInt8, kNCHW, fp32_out);",
             "USE_LITE_KERNEL(depthwise_conv2d, kARM, kInt8, kNCHW, fp32_out);\n"
             "#ifdef LITE_WITH_HIP\n"
             "USE_LITE_KERNEL(conv2d, kHIP, kInt8, kNCHW, int8_out);\nUSE_LITE_KERNEL(conv2d, kHIP, kInt8, kNCHW, fp32_out);\n"
             "USE_LITE_KERNEL(depthwise_conv2d, kHIP, kInt8, kNCHW, int8_out);\nUSE_LITE_KERNEL(depthwise_conv2d, kHIP, kInt8, kNCHW, fp32_out);\n"
             "USE_LITE_KERNEL(fc, kHIP, kInt8, kNCHW, int8out);\nUSE_LITE_KERNEL(fc, kHIP, kInt8, kNCHW, fp32out);\n"
             "USE_LITE_KERNEL(calib, kHIP, kInt8, kNCHW, fp32_to_int8);\nUSE_LITE_KERNEL(calib, kHIP, kInt8, kNCHW, int8_to_fp32);\n"
             "USE_LITE_KERNEL(io_copy, kHIP, kAny, kAny, host_to_device);\nUSE_LITE_KERNEL(io_copy, kHIP, kAny, kAny, device_to_host);\n"
             "USE_LITE_KERNEL(pool2d, kHIP, kFloat, kNCHW, def);\nUSE_LITE_KERNEL(pool2d, kHIP, kInt8, kNCHW, def);\n"
             "USE_LITE_KERNEL(elementwise_add, kHIP, kFloat, kNCHW, def);\nUSE_LITE_KERNEL(fusion_elementwise_add_activation, kHIP, kFloat, kNCHW, def);\n"
             "USE_LITE_KERNEL(softmax, kHIP, kFloat, kNCHW, def);\n"
             "#endif  // LITE_WITH_HIP"),
        ],
        "lite/kernels/CMakeLists.txt": [
            ("add_subdirectory(cuda)\n", "add_subdirectory(cuda)\nadd_subdirectory(hip)\n"),
        ],
        "lite/backends/CMakeLists.txt": [
            ("add_subdirectory(cuda)\n", "add_subdirectory(cuda)\nadd_subdirectory(hip)\n"),
        ],
    },
    "0006-mir-hip-conv-tail-fuse-pass.patch": {
        "lite/core/optimizer.h": [
            ("         \"runtime_context_assign_pass\",\n         \"argument_type_display_pass\",\n         \"lite_reshape_fuse_pass\",",
             "         \"hip_conv_tail_fuse_pass\",  // kHIP: conv + residual add (+relu) + calib, conv + max pool + calib\n"
             "         \"runtime_context_assign_pass\",\n         \"argument_type_display_pass\",\n         \"lite_reshape_fuse_pass\","),
        ],
        "lite/api/paddle_use_passes.h": [
            ("USE_MIR_PASS(lite_conv_elementwise_fuse_pass);\n", "USE_MIR_PASS(lite_conv_elementwise_fuse_pass);\nUSE_MIR_PASS(hip_conv_tail_fuse_pass);\n"),
        ],
        "lite/core/mir/CMakeLists.txt": [
            ("      fusion/conv_elementwise_fuse_pass.cc\n", "      fusion/conv_elementwise_fuse_pass.cc\n      fusion/hip_conv_tail_fuse_pass.cc\n"),
        ],
    },
}


NEW_FILES = {
    "0006-mir-hip-conv-tail-fuse-pass.patch": {
        # the matcher is THIS repository's file, verbatim (tests/test_patches.py compares the bytes)
        "lite/core/mir/fusion/hip_conv_tail_matcher.h": open(os.path.join(ROOT, "paddle-lite_amd", "lite", "core", "mir", "fusion", "hip_conv_tail_matcher.h")).read(),
        "lite/core/mir/fusion/hip_conv_tail_fuse_pass.h": PASS_H,
        "lite/core/mir/fusion/hip_conv_tail_fuse_pass.cc": PASS_CC,
    },
    "0005-runtime-sync-hooks-copysync-kernel-list-and-cmake-for-kHIP.patch": {
        "lite/kernels/hip/CMakeLists.txt":
            "if((NOT LITE_ON_MODEL_OPTIMIZE_TOOL) AND (NOT LITE_WITH_PYTHON) AND (NOT LITE_WITH_HIP))\n    return()\nendif()\n\n"
            "message(STATUS \"compile with lite HIP (gfx950) kernels\")\n\n"
            "# kernel classes of this repository (paddle-lite_amd/lite/kernels/hip): plain C++ over the C ABI of libplhip.so\n"
            "add_kernel(conv2d_hip HIP basic SRCS conv_compute.cc DEPS ${lite_kernel_deps} target_wrapper_hip plhip)\n"
            "add_kernel(fc_compute_hip HIP basic SRCS fc_compute.cc DEPS ${lite_kernel_deps} target_wrapper_hip plhip)\n"
            "add_kernel(glue_compute_hip HIP basic SRCS glue_compute.cc DEPS ${lite_kernel_deps} target_wrapper_hip plhip)\n",
        "lite/backends/hip/CMakeLists.txt":
            "if(NOT LITE_WITH_HIP)\n    return()\nendif()\n\n"
            "# libplhip.so: the gfx950 kernels + C ABI (include/plhip.h), built by hipcc (paddle-lite_amd/csrc/Makefile)\n"
            "add_library(plhip SHARED IMPORTED GLOBAL)\n"
            "set_property(TARGET plhip PROPERTY IMPORTED_LOCATION ${PLHIP_ROOT}/libplhip.so)\n"
            "include_directories(${PLHIP_ROOT}/../include)\n"
            "lite_cc_library(target_wrapper_hip SRCS target_wrapper.cc DEPS plhip)\n",
    },
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    a = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "patches"), exist_ok=True)
    for pname, files in EDITS.items():
        out = []
        for rel, edits in files.items():
            src = open(os.path.join(a.reference, rel)).read()
            dst = src
            for old, new in edits:
                assert dst.count(old) >= 1, "%s: anchor not found:\n%s" % (rel, old)
                dst = dst.replace(old, new, 1)
            diff = difflib.unified_diff(src.splitlines(True), dst.splitlines(True), "a/" + rel, "b/" + rel, n=3)
            # a last line without a newline (lite/api/paddle_use_kernels.h) needs git's marker
            out.append("diff --git a/%s b/%s\n" % (rel, rel) + "".join(l if l.endswith("\n") else l + "\n\\ No newline at end of file\n" for l in diff))
        for rel, text in NEW_FILES.get(pname, {}).items():
            lines = text.splitlines(True)
            out.append("diff --git a/%s b/%s\nnew file mode 100644\n--- /dev/null\n+++ b/%s\n@@ -0,0 +1,%d @@\n" % (rel, rel, rel, len(lines)) +
                       "".join("+" + l for l in lines))
        with open(os.path.join(ROOT, "patches", pname), "w") as f:
            f.write("".join(out))
        print("wrote patches/%s (%d files)" % (pname, len(files)))


if __name__ == "__main__":
    main()
