"""numpy-only counterpart of the reference's simulator side (Simulator/simulator/run_simulator.py,
Tests/run_test.py) for driving a planner process over the two FIFOs -- what `run_test.py` does with
cv2 and matplotlib, reduced to what the planner sees: the map, the circular "field of view"
patches, the heuristic hint, and the messages of the wire protocol.

cv2 is not a dependency: the Gaussian blur, the filled circle and the elliptic dilation are
restated on numpy arrays, following OpenCV's definitions (fixed-point Gaussian kernel with reflect-101
borders, x^2 + y^2 <= r^2 disc -- what cv2.circle fills for the radii used here --, 3x3 ellipse = cross).
Blur, inversion, penalty and reveal are pinned by the reference's own recorded mission log (see gaussian_blur);
the dilation is not (that mission ran with a C-space diameter of 1).
"""
import errno
import os
import struct
import subprocess
import time

import numpy as np


# ---- map preparation (run_simulator.py:106-113, run_test.py:98-104) -------------------------
def gaussian_kernel_fixed(ksize):
    """cv2.GaussianBlur(img, (k, k), 0) on 8-bit images: the kernel OpenCV >= 4.1 applies in fixed point -- the Gaussian of
    sigma = 0.3 * ((k - 1) / 2 - 1) + 0.8 (a fixed table for k <= 7) in 8 fractional bits, the rounding error of each coefficient carried
    to the next one and the centre taking what is left of 256.  k = 13: [1 5 10 19 30 41 44 41 30 19 10 5 1]."""
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if ksize in small:
        k = np.array(small[ksize], np.float64)
    else:
        sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
        x = np.arange(ksize) - (ksize - 1) * 0.5
        k = np.exp(-0.5 / (sigma * sigma) * x * x)
        k = k / k.sum()
    out = np.zeros(ksize, np.int64)
    err, acc = 0.0, 0
    for i in range(ksize // 2):
        adj = k[i] * 256 + err
        v = int(np.rint(adj))
        err = adj - v
        out[i] = out[ksize - 1 - i] = v
        acc += v
    out[ksize // 2] = 256 - 2 * acc
    return out


def gaussian_blur(img, ksize=3):
    """separable, reflect-101 borders, both passes exact in integers, one rounding at the end ((v + 2^15) >> 16).
    With this blur, the penalty and the reveal radius of Simulator/simulator/run_simulator.py:147,171 the reference's recorded
    mission (Tests/Results/noise-trap/planner_opt0.log) is reproduced to the last printed digit for all its 134 steps
    (tests/test_reference_mission.py) -- which is what pins this restatement to cv2, not a claim about cv2 in general."""
    if ksize <= 1:
        return img.copy()
    k, r = gaussian_kernel_fixed(ksize), ksize // 2
    a = np.pad(img.astype(np.int64), r, mode="reflect")
    h = sum(k[i] * a[:, i:i + img.shape[1]] for i in range(ksize))
    v = sum(k[i] * h[i:i + img.shape[0], :] for i in range(ksize))
    return np.clip((v + 32768) >> 16, 0, 255).astype(np.uint8)


def gaussian_blur3(img):
    """3x3 Gaussian as cv2.GaussianBlur(img, (3, 3), 0): [1 2 1]/4"""
    return gaussian_blur(img, 3)


def simulation_data(img_h, low_res_penalty=10, filter_size=3):
    """(low-resolution costs, high-resolution costs) of a grey-scale bitmap: cost = ~pixel, 0 -> 1;
    the low-resolution map is the blurred bitmap plus a saturating penalty (run_test.py: 3 / 10; run_simulator.py: 13 / 15)"""
    h = (~img_h).astype(np.uint8)
    h = h + (h == 0)
    l = (~gaussian_blur(img_h, filter_size)).astype(np.uint8)
    l = l + (l == 0)
    l = np.minimum(l.astype(np.int32) + low_res_penalty, 255).astype(np.uint8)
    return l, h


def dilate(img, diameter):
    """grey-scale dilation by an elliptic structuring element of the given diameter (C-space)"""
    if diameter <= 1:
        return img.copy()
    r = diameter // 2
    yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
    mask = (xx * xx) / max(r * r, 1) + (yy * yy) / max(r * r, 1) <= 1.0
    if diameter == 3:
        mask = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)
    p = np.pad(img, r, mode="edge")
    out = np.zeros_like(img)
    for dy in range(2 * r + 1):
        for dx in range(2 * r + 1):
            if mask[dy, dx]:
                out = np.maximum(out, p[dy:dy + img.shape[0], dx:dx + img.shape[1]])
    return out


def round_patch_update(data_l, data_h, center, radius):
    """Reveal the high-resolution data inside the disc around center = (col, row); returns
    (new low-resolution map, (top, left), (row slice, col slice)) -- the rectangle that bounds the
    disc, clipped at the top / left border only, as the reference does (run_simulator.py:9-28)."""
    cx, cy = center
    top, left = max(cy - radius, 0), max(cx - radius, 0)
    bottom, right = cy + radius + 1, cx + radius + 1
    yy, xx = np.ogrid[:data_l.shape[0], :data_l.shape[1]]
    disc = (xx - cx) ** 2 + (yy - cy) ** 2 <= radius * radius
    out = data_l.copy()
    out[disc] = data_h[disc]
    return out, (top, left), (slice(top, min(bottom, data_l.shape[0])), slice(left, min(right, data_l.shape[1])))


# ---- the simulator's end of the pipes (run_simulator.py:38-103) -----------------------------
class Pipes:
    def __init__(self, to_planner, from_planner, alive=lambda: True, timeout=60.0):
        # open order of run_test.py:92-93.  A FIFO's write end cannot be opened before the reader
        # has it open; poll instead of blocking, so that a planner that died at start-up is noticed
        t0 = time.time()
        while True:
            try:
                fd = os.open(to_planner, os.O_WRONLY | os.O_NONBLOCK)
                break
            except OSError as e:
                if e.errno != errno.ENXIO:
                    raise
                if not alive():
                    raise RuntimeError("the planner process exited before opening its pipes")
                if time.time() - t0 > timeout:
                    raise RuntimeError("the planner process did not open %s" % to_planner)
                time.sleep(0.01)
        os.set_blocking(fd, True)
        self.o = os.fdopen(fd, "wb")
        self.i = open(from_planner, "rb")

    def send(self, fmt, *v):
        self.o.write(struct.pack("<" + fmt, *v))

    def send_bytes(self, b):
        self.o.write(b)

    def flush(self):
        self.o.flush()

    def recv(self, fmt):
        n = struct.calcsize("<" + fmt)
        b = self.i.read(n)
        if len(b) != n:
            raise EOFError("planner closed the pipe")
        return struct.unpack("<" + fmt, b)

    def recv_bytes(self, n):
        b = self.i.read(n)
        if len(b) != n:
            raise EOFError("planner closed the pipe")
        return b

    def close(self):
        self.o.close()
        self.i.close()


def run_mission(cmd, pipe_to_planner, pipe_from_planner, img_h, start, goal, radius=5, cspace_diameter=1,
                low_res_penalty=10, use_heuristic=False, max_moves=10000, on_map=None, on_move=None, display_shift=0.0,
                append_pipes=True):
    """One mission as Tests/run_test.py:85-177 runs it: launch the planner process `cmd`, send the
    C-space of the low-resolution map, then per robot position reveal the disc of radius `radius`,
    send its bounding patch and the heuristic hint, receive the planned path.  start / goal are
    (row, col) positions and go in-band (the DFM driver's protocol; a driver with the 11-argument
    form ignores nothing because none is sent then -- pass start=None).
    on_map(cspace, min_cost) is called with what the planner receives first;
    on_move(i, position, top, left, patch, min_cost, (path, costs, dist, cost, times)) after every reply
    (position without `display_shift`, the half cell the DFM driver adds for display).  Returns the list of positions visited and whether the planner reported the end."""
    for p in (pipe_to_planner, pipe_from_planner):
        if not os.path.exists(p):
            os.mkfifo(p)
    if append_pipes:                    # the two-argument form: <fifo_in> <fifo_out> of the planner
        cmd = list(cmd) + [pipe_to_planner, pipe_from_planner]
    proc = subprocess.Popen(cmd, stdout=subprocess.DEVNULL)
    trace = []
    finished = False
    try:
        io = Pipes(pipe_to_planner, pipe_from_planner, alive=lambda: proc.poll() is None)
        assert io.recv("b") == (0,)
        io.send("b", 0)
        data_l, data_h = simulation_data(img_h, low_res_penalty)
        cspace = dilate(data_l, cspace_diameter)
        min_cost = int(cspace.min())
        height, width = cspace.shape
        io.send("ii", width, height)
        io.send_bytes(np.ascontiguousarray(cspace).tobytes())
        if start is not None:
            io.send("ffffB", float(start[0]), float(start[1]), float(goal[0]), float(goal[1]), 0)
        io.send("i", min_cost)
        io.flush()
        if on_map is not None:
            on_map(cspace.copy(), min_cost)
        prev = None
        while len(trace) < max_moves:
            (code,) = io.recv("b")
            if code == 2:
                finished = True
                break
            assert code == 1, code
            xr, yr, step_cost = io.recv("fff")
            x, y = xr - display_shift, yr - display_shift
            if prev == (x, y):          # run_test.py:135-139: the planner got stuck
                break
            prev = (x, y)
            trace.append((x, y))
            center = (int(round(yr)), int(round(xr)))         # (col, row) of the position as received, run_test.py:143
            data_l, (top, left), ranges = round_patch_update(data_l, data_h, center, radius)
            cspace = dilate(data_l, cspace_diameter)
            patch = np.ascontiguousarray(cspace[ranges[0], ranges[1]])
            if use_heuristic:
                min_cost = int(cspace.min())
            io.send("b", 1)
            io.send("iiii", top, left, patch.shape[0], patch.shape[1])
            io.send_bytes(patch.tobytes())
            io.send("i", min_cost)
            io.flush()
            assert io.recv("b") == (3,)
            (n,) = io.recv("i")
            path = np.array(io.recv("%df" % (2 * n)), np.float32).reshape(n, 2)
            costs = np.array(io.recv("%df" % max(n - 1, 0)), np.float32)     # one cost per segment (run_simulator.py:82-83)
            dist, cost = io.recv("ff")
            times = io.recv("fff")
            if on_move is not None:
                on_move(len(trace) - 1, (x, y), top, left, patch, min_cost, (path, costs, dist, cost, times))
        io.send("b", 2)
        io.flush()
        if finished:
            proc.wait(timeout=60)
        io.close()
    finally:
        if proc.poll() is None:
            proc.kill()
    return trace, finished
