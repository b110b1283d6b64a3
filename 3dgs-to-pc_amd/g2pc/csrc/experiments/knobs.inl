// EXPERIMENTS ONLY (-DG2PC_EXPERIMENTS; not part of libg2pc.so): the process-global tuning and diagnostic entry points of
// rounds 2-4.  tools/ and bench.py's tuning flags bind them from libg2pc_exp.so; the product ABI (include/g2pc.h) has none.
#pragma GCC visibility push(default)
extern "C" {
/* per-chunk walk statistics of the PY blends: u32[8 * num_chunks] (batch 1): [0] tile list length, [1] entries walked; the dual-list
 * kernel also [2] start and [3] duration on the 100 MHz wall clock, [4] HW_ID, [5] XCC_ID, [6] visits after the cull */
int g2pc_raster_debug_chunk_work(uint32_t* buf) { g2pc::g_knobs.chunk_work = buf; return G2PC_OK; }
/* n empty kernels after the preprocess of every python-semantics camera batch (what a kernel boundary costs a job) */
int g2pc_debug_set_extra_launches(int n) { g2pc::g_knobs.extra_launches = n > 0 ? n : 0; return G2PC_OK; }
/* threads per block (64, 128 or 256) of the python-semantics head kernels without block-level cooperation */
int g2pc_debug_set_head_threads(int threads) {
    if (threads != 64 && threads != 128 && threads != 256) return G2PC_ERR_ARG;
    g2pc::g_knobs.head_threads = threads;
    return G2PC_OK;
}
/* the dual-list PY blend stops every tile walk after `batches` 64-entry batches (0 = off).  The results are WRONG. */
int g2pc_debug_set_walk_cap(int batches) { g2pc::g_knobs.walk_cap = batches > 0 ? batches : 0; return G2PC_OK; }
/* blend kernel for 2 sub-blocks per wave: see Knobs::blend_variant */
int g2pc_set_blend_variant(int variant) { g2pc::g_knobs.blend_variant = variant; return G2PC_OK; }
/* depth order of the capture-safe camera call: 1 = bucket sort (the product's), 0 = 4-pass radix.  Identical results. */
int g2pc_set_depth_sort(int bucket) { g2pc::g_knobs.depth_bucket_sort = bucket ? 1 : 0; return G2PC_OK; }
/* a HIP stream restricted to the CUs of `mask` (bit i of word i / 32 = CU i; hipExtStreamCreateWithCUMask): the experiment
 * "heads and blends on disjoint CUs" of DESIGN.md §5.3 (the emulator has no such thing: a plain handle) */
int g2pc_debug_stream_create_cu_mask(const uint32_t* mask, uint32_t words, void** stream) {
    if (!mask || !words || !stream) return G2PC_ERR_ARG;
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask);
    if (e != hipSuccess) { g2pc::set_error("stream_create_cu_mask", hipGetErrorString(e)); return G2PC_ERR_LAUNCH; }
    *stream = (void*)s;
    return G2PC_OK;
}
}
#pragma GCC visibility pop
