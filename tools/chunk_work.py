"""Diagnostics: distribution of blend work per chunk for a few cameras of the bench scene (run on the GPU box).

Per camera: list lengths / entries walked per chunk, and -- from the per-wave clocks the dual-list kernel records -- how
long each single-wave block ran, how many were resident over time, and what the launch would take if its waves were
spread perfectly (sum of wave time / slots) against what it took."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd")]
import numpy as np, torch
import gauss_render, camera_handler
from gauss_handler import Gaussians
from g2pc import _native as nv
from g2pc.synth import make_scene, make_cameras

dev = "cuda:0"
NCAM = int(os.environ.get("CHUNK_WORK_CAMERAS", "6"))
sc = make_scene(1_000_000, 1237, device=dev)
G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
tr, intr = make_cameras(50)
# one camera at a time: the clocks of one launch alone on the device.  CHUNK_WORK_PIPELINE=1: through the batched camera call
# (the per-camera blend plan: longest-list-first order, hand-over of long walks), flushed after every camera; otherwise the
# two-call path (static chunk order, no hand-over)
PIPE = os.environ.get("CHUNK_WORK_PIPELINE", "0") == "1"
gauss_render.PIPELINE_STREAMS = 2 if PIPE else 1
gauss_render.CAMERA_BATCH = 1
R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
names = sorted(tr)[:NCAM]
q = lambda a: [int(np.percentile(a, p)) for p in (50, 90, 99, 99.9, 100)]
out = []
for rep in range(2):                                  # the first pass warms the process up
    for name in names:
        cam = camera_handler.get_camera("python", torch.tensor(tr[name]), intr[name], colour_resolution=1280)
        lay = R._layout(cam.image_width, cam.image_height)
        nchunks = lay.c.num_chunks
        buf = torch.zeros(8 * nchunks, dtype=torch.int32, device=dev)
        nv.experiments().g2pc_raster_debug_chunk_work(nv.ptr(buf))
        torch.cuda.synchronize()
        nv.PROFILE = {}
        R(cam, return_image=False)
        R.flush()
        torch.cuda.synchronize()
        prof = {k: round(v[1], 3) for k, v in nv.profile_summary().items()}
        nv.PROFILE = None
        if rep == 0:
            continue
        w = buf.cpu().numpy().astype(np.int64).reshape(-1, 8) & 0xFFFFFFFF
        ln, done, t0, dur, hw, xcc, vis = (w[:, i] for i in range(7))
        life = w[:, 7] >> 16                           # main walk + the exported quarters this wave ran afterwards (10 ns ticks)
        exported, quarters = int((w[:, 7] & 1).sum()), int(((w[:, 7] >> 8) & 0xFF).sum())
        own = dur.copy()
        dur = np.maximum(dur, life)
        full = (done >= ln) & (ln > 0)
        print(name, "chunks", nchunks, "list len p50/90/99/99.9/max", q(ln), "walked", q(done), "sum walked %.3g" % done.sum(),
              "walked-full chunks", int(full.sum()), prof)
        t0 = (t0 - t0.min()) & 0xFFFFFFFF             # 100 MHz ticks = 10 ns
        t1 = t0 + dur
        span = float(t1.max()) * 0.01                  # us
        wave_us = float(dur.sum()) * 0.01
        simd = ((xcc & 0xF) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 8) & 0xF) << 4) | ((hw >> 4) & 3)
        nsimd = len(np.unique(simd))
        # residency over time: waves in flight at 64 sample points
        ts = np.linspace(0, t1.max(), 65)[:-1] + t1.max() / 128.0
        res = [(int(((t0 <= t) & (t1 > t)).sum())) for t in ts]
        order = np.argsort(-dur)
        rec = {"camera": name, "span_us": round(span, 1), "sum_wave_us": round(wave_us, 1), "simds_seen": nsimd,
               "exported_chunks": exported, "quarters_run": quarters, "sum_own_walk_us": round(float(own.sum()) * 0.01, 1),
               "own_walk_us p50/90/99/max": [round(float(np.percentile(own, p)) * 0.01, 1) for p in (50, 90, 99, 100)],
               "mean_resident_waves_per_simd": round(wave_us / span / max(nsimd, 1), 2),
               "dur_us p50/90/99/99.9/max": [round(x * 0.01, 1) for x in q(dur)],
               "start_us p50/90/99/max": [round(float(np.percentile(t0, p)) * 0.01, 1) for p in (50, 90, 99, 100)],
               "residency_64_samples": res,
               "longest10": [{"dur_us": round(float(dur[i]) * 0.01, 1), "start_us": round(float(t0[i]) * 0.01, 1),
                              "walked": int(done[i]), "len": int(ln[i]), "visits": int(vis[i])} for i in order[:10]],
               "us_per_walked_entry_by_decile_of_walk": [
                   round(float(dur[s].sum() * 0.01 / max(done[s].sum(), 1)), 4)
                   for s in np.array_split(np.argsort(done), 10)],
               "visits": int(vis.sum()), "visits_per_walked_entry_by_decile_of_walk": [
                   round(float(vis[s].sum() / max(done[s].sum(), 1)), 3) for s in np.array_split(np.argsort(done), 10)],
               "cycles_per_visit_by_decile_of_walk (2.4 GHz, per wave)": [
                   round(float(dur[s].sum() * 24.0 / max(vis[s].sum(), 1)), 1) for s in np.array_split(np.argsort(done), 10)],
               "waves_per_xcd": np.bincount((xcc & 0xF).astype(np.int64), minlength=8).tolist(),
               "blend_region_ms": prof.get("raster_blend")}
        print(json.dumps(rec))
        out.append(rec)
nv.experiments().g2pc_raster_debug_chunk_work(None)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", os.environ.get("CHUNK_WORK_OUT", "chunk_clocks.json")), "w"), indent=1)
