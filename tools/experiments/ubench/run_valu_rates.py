"""GPU box: cycles per wave-instruction and SIMD for the blend's instruction kinds (tools/experiments/ubench/valu_rates.hip,
built by tools/experiments/ubench/build.sh)."""
import ctypes as C, os, json, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "libubench.so"))
lib.ubench_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(4, device=dev)
names = {0: "v_fma_f32", 1: "v_pk_fma_f32", 2: "v_exp_f32+v_fma_f32", 3: "v_pk_mul_f32", 4: "v_pk_add_f32", 5: "v_mul_f32+v_min_f32",
         6: "v_exp_f32", 7: "ds_read_b128 broadcast"}
insts_per_trip = {0: 8, 1: 8, 2: 16, 3: 8, 4: 8, 5: 16, 6: 8, 7: 8}
res = {}
iters = 20000
for waves_per_simd in (1, 2, 4, 8):
    blocks = 256 * 4 * waves_per_simd
    for kind, name in names.items():
        st = torch.cuda.current_stream().cuda_stream
        lib.ubench_launch(kind, blocks, 100, C.c_void_p(out.data_ptr()), C.c_void_p(st))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.ubench_launch(kind, blocks, iters, C.c_void_p(out.data_ptr()), C.c_void_p(st))
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        wave_insts_per_simd = waves_per_simd * iters * insts_per_trip[kind]
        res["%s @%d waves/SIMD" % (name, waves_per_simd)] = {"ms": ms, "ns_per_wave_inst_per_simd": ms * 1e6 / wave_insts_per_simd,
                                                            "cycles_at_2.4GHz": ms * 1e6 / wave_insts_per_simd * 2.4}
print(json.dumps(res, indent=1))
