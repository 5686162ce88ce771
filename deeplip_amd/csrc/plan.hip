// Step plans: a recorded sequence of dlip_* launches replayed with one call (include/deeplip_hip.h).
//
// The hot path is ~45 short kernels per step; launched one by one from the Python host the GPU idles between
// them (round 1: 1.4 ms of a 6.5 ms step).  A plan is the HIP-native answer: every dlip_* entry point launches
// on the caller's stream, so the host records a step ONCE by running it between dlip_plan_begin and
// dlip_plan_end (HIP stream capture, thread-local mode) and then replays the instantiated hipGraph with
// dlip_plan_run -- one host call per step, kernel boundaries back to back on the device.  The library adds
// nothing to the graph beyond what the caller launched; buffers stay caller-owned (the Python binding keeps
// every tensor of the recorded step alive in an arena, deeplip_amd/plan.py).
#include "dlip_common.h"

#include <vector>

namespace {

struct Plan {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int device = 0;
  int kernel_nodes = 0;
  int nodes = 0;
};

}  // namespace

extern "C" int dlip_plan_begin(dlip_stream_t stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  DLIP_CHECK_ARG(st != nullptr);   // the null stream cannot be captured
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  hipError_t e = hipStreamIsCapturing(st, &status);
  if (e != hipSuccess) return (int)e;
  DLIP_CHECK_ARG(status == hipStreamCaptureStatusNone);
  e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  return e == hipSuccess ? DLIP_OK : (int)e;
}

extern "C" int dlip_plan_end(dlip_stream_t stream, dlip_plan_t* plan) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  DLIP_CHECK_ARG(st != nullptr && plan != nullptr);
  *plan = nullptr;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(st, &graph);
  if (e != hipSuccess) return (int)e;
  DLIP_CHECK_ARG(graph != nullptr);
  Plan* p = new Plan();
  p->graph = graph;
  (void)hipGetDevice(&p->device);
  size_t n = 0;
  if (hipGraphGetNodes(graph, nullptr, &n) == hipSuccess && n > 0) {
    std::vector<hipGraphNode_t> nodes(n);
    if (hipGraphGetNodes(graph, nodes.data(), &n) == hipSuccess) {
      p->nodes = (int)n;
      for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++p->kernel_nodes;
      }
    }
  }
  e = hipGraphInstantiate(&p->exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    delete p;
    return (int)e;
  }
  *plan = p;
  return DLIP_OK;
}

extern "C" int dlip_plan_run(dlip_plan_t plan, dlip_stream_t stream) {
  DLIP_CHECK_ARG(plan != nullptr);
  Plan* p = static_cast<Plan*>(plan);
  hipError_t e = hipGraphLaunch(p->exec, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DLIP_OK : (int)e;
}

extern "C" int dlip_plan_launches(dlip_plan_t plan) {
  if (plan == nullptr) return DLIP_EINVAL;
  return static_cast<Plan*>(plan)->kernel_nodes;
}

extern "C" int dlip_plan_destroy(dlip_plan_t plan) {
  if (plan == nullptr) return DLIP_OK;
  Plan* p = static_cast<Plan*>(plan);
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  delete p;
  return DLIP_OK;
}
