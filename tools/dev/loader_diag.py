import sys, os, tempfile, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multiprocessing as mp

def decode_only(files):
  from oatomobile_amd import replay
  t0 = time.perf_counter()
  for f in files: replay.load_datum(f)
  return time.perf_counter() - t0

def zlib_only(n):
  import zlib
  rng = np.random.default_rng(0)
  raw = ((rng.random((200, 200, 2)) < 0.12) * 0.4).astype(np.float32).tobytes()
  c = zlib.compress(raw)
  t0 = time.perf_counter()
  for _ in range(n): zlib.decompress(c)
  return time.perf_counter() - t0

if __name__ == "__main__":
  from oatomobile_amd import replay
  from bench import synth_batch
  d = tempfile.mkdtemp()
  ep = replay.Episode(d, "ep")
  rng = np.random.default_rng(0)
  lidar, vec, goal = synth_batch(rng, 256, 2)
  for i in range(1024):
    ep.append(lidar=lidar[i % 256], velocity=vec[i % 256, :3], is_at_traffic_light=vec[i % 256, 3], traffic_light_state=vec[i % 256, 4],
              player_future=np.zeros((80, 3), np.float32))
  files = ep.files()
  for W in (1, 8, 32, 64):
    with mp.get_context("spawn").Pool(W) as pool:
      pool.map(zlib_only, [1] * W)  # warm up (imports)
      t0 = time.perf_counter(); ts = pool.map(decode_only, [files[i::W] for i in range(W)]); wall = time.perf_counter() - t0
      t0 = time.perf_counter(); tz = pool.map(zlib_only, [200] * W); wallz = time.perf_counter() - t0
      print("W=%2d decode distinct files: %.0f obs/s (per-worker busy %.2f s of wall %.2f) | zlib only: %.0f /s" % (
          W, len(files) / wall, float(np.mean(ts)), wall, 200 * W / wallz))
