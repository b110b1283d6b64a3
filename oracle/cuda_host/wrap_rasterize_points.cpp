// TEST INFRASTRUCTURE ONLY (oracle) -- compiles the reference's rasterize_points.cu (the torch binding of its CUDA
// rasteriser) for the host, from where it lies under /root/reference (path given by the build recipe, oracle/build_ref.py).
// The reference allocates its three scratch tensors on torch::kCUDA (rasterize_points.cu:92-96); this container has no
// GPU, so for THIS translation unit the name kCUDA is read as kCPU.  torch's own headers are included first and are
// therefore unaffected (they are include-guarded), the reference's text is not edited.
#include <torch/extension.h>
#define kCUDA kCPU
#include G2PC_REF_RASTERIZE_POINTS_CU
