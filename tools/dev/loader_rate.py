"""GPU box: where the multi-process replay spends its time (loader alone, pageable / registered H2D)."""
import sys, os, tempfile, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if __name__ == "__main__":
  from oatomobile_amd import replay
  from bench import synth_batch
  print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
  d = tempfile.mkdtemp()
  ep = replay.Episode(d, "ep")
  rng = np.random.default_rng(0)
  lidar, vec, goal = synth_batch(rng, 128, 2)
  for i in range(128):
    ep.append(lidar=lidar[i], velocity=vec[i, :3], is_at_traffic_light=vec[i, 3], traffic_light_state=vec[i, 4],
              player_future=np.cumsum(np.abs(rng.normal(size=(80, 3))), 0).astype(np.float32))
  files = ep.files() * 32
  for W in (8, 16, 32, 64):
    t0, n = None, 0
    for b in replay.DatumBatches(files, 512, workers=W):
      if t0 is None:
        t0 = time.time()
      else:
        n += b[0].shape[0]
    print("workers", W, "loader alone: %.0f obs/s" % (n / (time.time() - t0)))
  if torch.cuda.is_available():
    x = torch.empty(512, 200, 200, 2)
    xp = torch.empty(512, 200, 200, 2).pin_memory()
    for name, t in (("pageable", x), ("pinned", xp)):
      torch.cuda.synchronize(); t0 = time.time()
      for _ in range(5): y = t.to("cuda", non_blocking=True)
      torch.cuda.synchronize()
      print(name, "H2D of 164 MB: %.1f ms" % ((time.time() - t0) / 5 * 1e3))
