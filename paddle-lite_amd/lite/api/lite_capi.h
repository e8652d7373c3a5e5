/* lite_capi.h — a small C veneer over the C++ kernel classes (lite/kernels/hip) and the mini predictor, so that
 * the Python test-suite and bench.py can drive the SAME objects a Paddle-Lite build would (KernelFactory lookup,
 * SetContext / SetParam / PrepareForRun / Launch) through ctypes.  Test / bench plumbing only; the drop-in ABI of
 * the device code is include/plhip.h.  Every function returns 0 or -1 (message in pllite_last_error()). */
#ifndef PLLITE_CAPI_H_
#define PLLITE_CAPI_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* pllite_last_error(void);
/* Number of kernels registered for (op_type, kHIP, precision, layout) — registry smoke check. */
int pllite_registered_kernels(const char* op_type, int precision, int layout);
/* Adopt an external hipStream_t (e.g. torch's current stream) as the calling thread's default execution stream on
 * `device`.  A predictor binds the default execution state (stream + workspace) of the thread that CREATES it and keeps
 * it for life: pllite_run / pllite_run_instruction may be called from any thread (one at a time per predictor — a
 * predictor is single-threaded like the reference's, program.cc:282-306) and always enqueue on that stream. */
int pllite_adopt_stream(int device, void* stream);

typedef struct pllite_predictor pllite_predictor;
pllite_predictor* pllite_predictor_create(int device);
void pllite_predictor_destroy(pllite_predictor* p);
int pllite_add_feed(pllite_predictor* p, const char* name, const int64_t* dims, int ndims, int precision);
int pllite_add_io_copy(pllite_predictor* p, const char* in, const char* out, int host_to_device);
int pllite_add_calib(pllite_predictor* p, const char* in, const char* out, float scale, int fp32_to_int8);
int pllite_add_conv(pllite_predictor* p, const char* op_type, const char* in, const char* out, const int8_t* w,
                    const int64_t* w_dims, const float* bias, const int* strides, const int* paddings, int n_paddings,
                    const int* dilations, int groups, int act, float act_coef, float input_scale,
                    const float* weight_scale, int n_weight_scale, float output_scale, int int8_out,
                    const char* padding_algorithm);
int pllite_add_fc(pllite_predictor* p, const char* in, const char* out, const int8_t* w, int k, int n, const float* bias,
                  float input_scale, const float* weight_scale, int n_weight_scale, float output_scale, int int8_out,
                  int relu);
int pllite_add_global_avg_pool(pllite_predictor* p, const char* in, const char* out);
int pllite_add_softmax(pllite_predictor* p, const char* in, const char* out);
int pllite_add_pool(pllite_predictor* p, const char* in, const char* out, const char* pooling_type, const int* ksize,
                    const int* strides, const int* paddings4, int global_pooling, int exclusive, int ceil_mode);
int pllite_add_elementwise_add(pllite_predictor* p, const char* x, const char* y, const char* out, const char* act_type);

/* ---- graph mode (lite/api/graph_builder.h): ops as the optimiser sees them after its fusion passes; kernel choice
 * (int8_out / fp32_out), io_copy and calib placement are decided by pllite_graph_lower() with the reference's rules.
 * One graph per predictor; ops in topological order. ---- */
int pllite_graph_feed(pllite_predictor* p, const char* name, const int64_t* dims, int ndims, int precision);
int pllite_graph_conv(pllite_predictor* p, const char* op_type, const char* in, const char* out, const int8_t* w,
                      const int64_t* w_dims, const float* bias, const int* strides, const int* paddings, int n_paddings,
                      const int* dilations, int groups, int act, float act_coef, float input_scale,
                      const float* weight_scale, int n_weight_scale, const char* padding_algorithm);
int pllite_graph_fc(pllite_predictor* p, const char* in, const char* out, const int8_t* w, int k, int n, const float* bias,
                    float input_scale, const float* weight_scale, int n_weight_scale, int relu);
int pllite_graph_pool(pllite_predictor* p, const char* in, const char* out, const char* pooling_type, const int* ksize,
                      const int* strides, const int* paddings4, int global_pooling, int exclusive, int ceil_mode);
int pllite_graph_elementwise_add(pllite_predictor* p, const char* x, const char* y, const char* out, const char* act_type);
int pllite_graph_softmax(pllite_predictor* p, const char* in, const char* out);
int pllite_graph_fetch(pllite_predictor* p, const char* name);
/* kHIP graph-level fusions (graph_builder.h set_fuse): on by default; 0 = the reference program instruction for instruction. */
int pllite_graph_set_fuse(pllite_predictor* p, int on);
// opt-in: depthwise_conv2d[int8_out] -> sole consumer conv2d 1x1 as ONE instruction (GraphBuilder::set_fuse_dwpw)
int pllite_graph_set_fuse_dwpw(pllite_predictor* p, int on);
/* '\n'-separated plan (GraphBuilder::Plan) — needs no device. */
int pllite_graph_plan(pllite_predictor* p, char* buf, int cap);
/* Emit the program into the predictor; '\n'-separated host names of the fetched variables in buf. */
int pllite_graph_lower(pllite_predictor* p, char* buf, int cap);
/* A predictor object that can only plan (no device needed): for CPU tests of the lowering rules. */
pllite_predictor* pllite_predictor_create_planner(void);

/* ---- model ingestion (lite/model_parser/hip_model.h): parses a PLHIPM01 container, applies the reference's
 * quant/dequant, conv+bn, conv+activation, fc and elementwise+activation fusion semantics and fills the predictor's
 * graph (then: pllite_graph_plan / pllite_graph_lower).  Needs no device. ---- */
int pllite_load_model(pllite_predictor* p, const void* bytes, int64_t nbytes, int batch);
/* The fused parameters of graph op `index` (conv2d / depthwise_conv2d / fc) for inspection: element counts through
 * n_w / n_bias / n_scale; arrays are copied when the pointers are non-null (capacity = the counts of a first call). */
int pllite_graph_num_ops(pllite_predictor* p);
int pllite_graph_op_params(pllite_predictor* p, int index, char* type, int type_cap, int8_t* w, int64_t* n_w, float* bias,
                           int* n_bias, float* weight_scale, int* n_scale, float* input_scale, int* act, float* act_coef);

int pllite_set_input(pllite_predictor* p, const char* name, const void* host, int64_t bytes);
int pllite_run(pllite_predictor* p, int skip_io_copy);
// the device part of the program as one recorded launch graph (first call records; needs one earlier pllite_run)
int pllite_run_graph(pllite_predictor* p);
int pllite_sync(pllite_predictor* p);
/* Per-instruction stepping (bench.py brackets single launches with HIP events for the roofline object). */
int pllite_num_instructions(pllite_predictor* p);
int pllite_run_instruction(pllite_predictor* p, int index);
/* Copies a variable (host or device resident) to `host`; returns bytes written through *bytes. */
int pllite_get_var(pllite_predictor* p, const char* name, void* host, int64_t capacity, int64_t* bytes,
                   int64_t* dims4, int* ndims);
/* Device pointer of a variable (for all_gather of logits etc.); 0 if it is not on the device. */
void* pllite_var_device_ptr(pllite_predictor* p, const char* name);
/* Asynchronous device-to-device copy of a device-resident variable into caller memory, on the predictor's stream. */
int pllite_copy_var_to_device(pllite_predictor* p, const char* name, void* dst_dev, int64_t bytes);
/* Times instruction `index` with profile::DeviceTimer<TargetType::kHIP> (lite/core/profile/timer.h): `reps` laps of
 * one launch each after one untimed launch; returns the average / minimum lap in ms and the kernel_func_name the kernel
 * reports through SetProfileRuntimeKernelInfo (lite/core/kernel.h:66-72). */
int pllite_time_instruction(pllite_predictor* p, int index, int reps, float* avg_ms, float* min_ms, char* func_name, int cap);
/* '\n'-separated "op:target/precision/layout/alias -> kernel_func_name" list of the program. */
int pllite_kernel_names(pllite_predictor* p, char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif
