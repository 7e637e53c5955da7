"""Datum decode and cache packing without torch (numpy only).

Imported by `replay.py`, and run as a SCRIPT by `replay.pack_cache`'s worker processes (`python _datum.py <job.json>`):
a packing worker must not pay `import torch` + the HIP library (2 s each) for 0.1 s of numpy work, and must not inherit
the parent's HIP context through a fork — so this file has no package-relative imports.

Reference formats: datum files `oatomobile/datasets/carla.py:107-164`; the goal derivation is SURVEY.md 8(d) config 5.
"""

import io
import json
import os
import sys
from typing import Mapping, Sequence

import numpy as np

MODALITIES = ("lidar", "velocity", "is_at_traffic_light", "traffic_light_state", "player_future")


def load_datum(fname: str, modalities: Sequence[str] = MODALITIES, mode: bool = False,
               dataformat: str = "HWC") -> Mapping[str, np.ndarray]:
  """datasets/carla.py:107-164: float32 casts, scalars -> 1-D, optional HWC->CHW, optional driving-mode label
  ({0 FORWARD, 1 STOP, 2 LEFT, 3 RIGHT} from the last future waypoint), `name` = path."""
  assert dataformat in ("HWC", "CHW")
  sample = {}
  with open(fname, "rb") as f:  # one read of the (small, compressed) file: the zip directory walk then costs no syscalls
    blob = io.BytesIO(f.read())
  with np.load(blob) as datum:
    for attr in modalities:
      v = np.atleast_1d(datum[attr]).astype(np.float32)
      if v.ndim == 3 and dataformat == "CHW":
        v = np.transpose(v, (2, 0, 1))
      sample[attr] = v
  if mode and "player_future" in sample:
    x_T, y_T = sample["player_future"][-1, :2]
    norm = np.linalg.norm([x_T, y_T])
    theta = np.degrees(np.arccos(x_T / (norm + 1e-3)))
    label = 1 if norm < 3 else (2 if theta > 15 else (3 if theta <= -15 else 0))
    sample["mode"] = np.atleast_1d(label).astype(np.float32)
  sample["name"] = fname
  return sample


def goal_from_future(player_future: np.ndarray, num_goals: int = 10, stride: int = 8) -> np.ndarray:
  """`player_future[stride-1::stride][:num_goals, :2]`, padded by repeating the last waypoint."""
  g = np.asarray(player_future, dtype=np.float32)[stride - 1::stride][:num_goals, :2]
  if g.shape[0] < num_goals:
    g = np.concatenate([g, np.repeat(g[-1:], num_goals - g.shape[0], axis=0)], axis=0)
  return g


def _fill_rows(files, j0, lidar, vec, goal, num_goals, goal_stride):
  """Decodes `files` into rows j0.. of the batch arrays (numpy views; shared memory in the worker processes)."""
  for j, f in enumerate(files, start=j0):
    d = load_datum(f)
    lidar[j] = d["lidar"]
    vec[j, :3] = d["velocity"].reshape(3)
    vec[j, 3] = float(d["is_at_traffic_light"].reshape(-1)[0])
    vec[j, 4] = float(d["traffic_light_state"].reshape(-1)[0])
    goal[j] = goal_from_future(d["player_future"], num_goals, goal_stride)
  return len(files)



def code_bev(bits: np.ndarray, table: np.ndarray):
  """uint8 codes of a BEV given as uint32 BIT PATTERNS against `table` (sorted distinct uint32 patterns), growing the
  table by the patterns it lacks.  Returns (codes, table).  Working on bit patterns keeps -0.0 and +0.0 apart (float
  comparison merges them) — `table.view(float32)[codes]` is the input bit for bit.  NaN patterns raise."""
  flat = bits.reshape(-1)
  if table.size:
    c = np.minimum(np.searchsorted(table, flat), table.size - 1)
    miss = table[c] != flat
    if miss.any():
      table = np.union1d(table, np.unique(flat[miss])).astype(np.uint32)
      c = np.searchsorted(table, flat)
      if not np.array_equal(table[c], flat):
        raise RuntimeError("pack_cache: table lookup does not reproduce the BEV")  # cannot happen
  else:
    table = np.unique(flat).astype(np.uint32)
    c = np.searchsorted(table, flat)
  if np.isnan(table.view(np.float32)).any():
    raise ValueError("pack_cache: NaN in a BEV")
  if table.size > 256:
    raise ValueError("pack_cache: more than 256 distinct BEV values (%d): not a clipped histogram" % table.size)
  return c.astype(np.uint8).reshape(bits.shape), table


def pack_span(files, i0, out_dir, shape, num_goals, goal_stride, chunk):
  """Packs `files` into rows i0.. of `<out_dir>/codes.npy` (an existing memmap of `shape`), one datum at a time through
  a reused frame buffer (a chunk-sized float32 staging array costs more in first-touch page faults than the decode),
  coded against the table of THIS span as known so far; when a datum brings a new value the rows of the current chunk
  are re-coded at once, so every chunk is consistent with one table.  Returns [(row0, rows, that table)], vec, goal:
  the parent unifies the tables and re-codes the chunks whose table differs from the final one."""
  n, H, W, C = shape
  codes = np.lib.format.open_memmap(os.path.join(out_dir, "codes.npy"), mode="r+")
  assert codes.shape == tuple(shape) and codes.dtype == np.uint8
  vec = np.empty((len(files), 5), np.float32)
  goal = np.empty((len(files), num_goals, 2), np.float32)
  table = np.empty((0,), np.uint32)
  frame = np.empty((1, H, W, C), np.float32)
  bits = frame.view(np.uint32)
  spans = []
  for j0 in range(0, len(files), chunk):
    part = files[j0:j0 + chunk]
    for j, f in enumerate(part, start=j0):
      _fill_rows([f], 0, frame, vec[j:j + 1], goal[j:j + 1], num_goals, goal_stride)
      c, grown = code_bev(bits, table)
      if grown.size != table.size and j > j0:  # a new value inside the chunk: its earlier rows move to the new table
        remap = np.searchsorted(grown, table).astype(np.uint8)
        codes[i0 + j0:i0 + j] = remap[codes[i0 + j0:i0 + j]]
      table = grown
      codes[i0 + j] = c[0]  # code_bev compared table[c] with the frame's bit patterns element by element
    spans.append((i0 + j0, len(part), table.copy()))
  codes.flush()
  del codes
  return spans, vec, goal


if __name__ == "__main__":  # worker process of replay.pack_cache: `python _datum.py job.json` -> job["result"] (.npz)
  with open(sys.argv[1]) as fh:
    job = json.load(fh)
  try:
    spans, vec, goal = pack_span(job["files"], job["i0"], job["out_dir"], tuple(job["shape"]), job["num_goals"],
                                 job["goal_stride"], job["chunk"])
    np.savez(job["result"], vec=vec, goal=goal, rows=np.array([(a, b) for a, b, _ in spans], np.int64),
             sizes=np.array([t.size for _, _, t in spans], np.int64),
             tables=np.concatenate([t for _, _, t in spans]) if spans else np.empty((0,), np.uint32))
  except ValueError as exc:  # refused data (NaN, > 256 values): the parent re-raises it as ValueError
    with open(job["result"] + ".err", "w") as fh:
      fh.write(str(exc))
    sys.exit(3)
