"""Seed-exact synthetic workloads: cost maps, obstacle
blocks, start/goal and the 100-patch replan script.  Pure integer / float64
numpy, so every language can reproduce the bytes."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def h64(seed, i, j):
    i = np.asarray(i, dtype=np.uint64)
    j = np.asarray(j, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return splitmix64(np.uint64(seed) ^ (i * np.uint64(0x9E3779B97F4A7C15)) ^ (j * np.uint64(0xC2B2AE3D27D4EB4F)))


def start_goal(width, length):
    """start (8,8) -> goal (length-8, width-8) in (x=row, y=col)."""
    return (8.0, 8.0), (float(length - 8), float(width - 8))


def cost_map(seed, width, length, obstacles=True):
    """uint8 [length][width]: bilinear x16 upsample of a hashed lattice into
    [1,200]; 8x8 obstacle blocks (255) with probability 1/64, cleared within 16
    cells of start and goal."""
    xi = np.arange(length, dtype=np.int64)
    yi = np.arange(width, dtype=np.int64)
    lx0, ly0 = xi // 16, yi // 16
    tx = (xi % 16).astype(np.float64) / 16.0
    ty = (yi % 16).astype(np.float64) / 16.0
    nlx, nly = length // 16 + 2, width // 16 + 2
    li, lj = np.meshgrid(np.arange(nlx), np.arange(nly), indexing="ij")
    lat = (h64(seed, li, lj) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    a = lat[lx0][:, ly0]
    b = lat[lx0][:, ly0 + 1]
    c = lat[lx0 + 1][:, ly0]
    d = lat[lx0 + 1][:, ly0 + 1]
    TX, TY = tx[:, None], ty[None, :]
    v = (a * (1 - TY) + b * TY) * (1 - TX) + (c * (1 - TY) + d * TY) * TX
    cost = (1 + np.floor(199.0 * v)).astype(np.uint8)
    if obstacles:
        nbx, nby = (length + 7) // 8, (width + 7) // 8
        bi, bj = np.meshgrid(np.arange(nbx), np.arange(nby), indexing="ij")
        obs_b = (h64(seed ^ 0xA5A5, bi, bj) % np.uint64(64)) == 0
        obs = np.repeat(np.repeat(obs_b, 8, axis=0), 8, axis=1)[:length, :width]
        (sx, sy), (gx, gy) = start_goal(width, length)
        nx = (np.abs(xi - sx) <= 16)[:, None] & (np.abs(yi - sy) <= 16)[None, :]
        ng = (np.abs(xi - gx) <= 16)[:, None] & (np.abs(yi - gy) <= 16)[None, :]
        cost[obs & ~(nx | ng)] = 255
    # C order: fancy indexing above leaves a column-major array, and consumers that hand out the raw
    # buffer (torch.from_numpy(...).data_ptr()) would see the transposed map
    return np.ascontiguousarray(cost)


def replan_script(seed, width, length, n_patches=100, size=31, stride=5):
    """Patches marching along the start->goal diagonal with the robot.
    Yields (k, start_xy, top, left, patch[size][size]) for k = 1..n_patches;
    patch bytes are 1 + h(seed^k, x, y) % 200 at global cell (x, y)."""
    (sx, sy), _ = start_goal(width, length)
    r = size // 2
    for k in range(1, n_patches + 1):
        px = int(min(sx + stride * k, length - 9))
        py = int(min(sy + stride * k, width - 9))
        top = max(0, min(px - r, length - size))
        left = max(0, min(py - r, width - size))
        XI, YI = np.meshgrid(np.arange(top, top + size), np.arange(left, left + size), indexing="ij")
        patch = (1 + (h64(seed ^ k, XI, YI) % np.uint64(200))).astype(np.uint8)
        yield k, (float(px), float(py)), top, left, patch
