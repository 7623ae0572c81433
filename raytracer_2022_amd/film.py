"""Row sharding of the film across ranks (the reference's thread scheme one level up).

The reference shuffles the row ids and hands each of its 8 threads a contiguous
section (raytracer/src/main.rs:93-116); each thread sends its Vec<Color> back once
(main.rs:157,171-183). Here the units are rows of an N-frame film strip — N frames
of the same scene and camera that differ only in their RNG key — dealt cyclically
from one shuffled list, so every rank renders the same number of rows drawn from
the same mix (weak scaling: per-GPU work is fixed as N grows). No collective is on
the data path; the only exchange is the final gather of the row buffers.
"""
import numpy as np

from .host import shuffled_rows


def strip_rows(height, n_frames, seed):
    """Shuffled global row ids of an n_frames-frame strip (ids address frame g // height, row g % height)."""
    return shuffled_rows(height * n_frames, seed)


def rank_rows(height, n_frames, seed, rank, world_size):
    """The rows rank `rank` renders: every world_size-th entry of the shuffled strip."""
    return np.ascontiguousarray(strip_rows(height, n_frames, seed)[rank::world_size])


def assemble(parts, rows_per_rank, height, n_frames, width):
    """Rank 0's side of the gather: place every rank's (n_rows, width, 3) sums into the
    (n_frames, height, width, 3) strip, row y stored at index y (y up, like main.rs:142)."""
    film = np.full((n_frames, height, width, 3), np.nan, dtype=np.float64)
    for part, rows in zip(parts, rows_per_rank):
        rows = np.asarray(rows, dtype=np.int64)
        film[rows // height, rows % height] = np.asarray(part).reshape(len(rows), width, 3)
    return film
