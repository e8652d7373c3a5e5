#!/usr/bin/env python3
"""Authoring tool of patches/*.patch: the kHIP touch-points inside a real Paddle-Lite tree.
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
            out.append("diff --git a/%s b/%s\n" % (rel, rel) + "".join(diff))
        with open(os.path.join(ROOT, "patches", pname), "w") as f:
            f.write("".join(out))
        print("wrote patches/%s (%d files)" % (pname, len(files)))


if __name__ == "__main__":
    main()
