"""CPU: the synthetic workload generator is seed-exact."""
import hashlib

import numpy as np

import ufm_amd

synth = ufm_amd.synth


def test_splitmix64_known_values():
    # splitmix64 of 0 and 1 (public test vectors of the generator)
    assert int(synth.splitmix64(np.uint64(0))) == 0xE220A8397B1DCDAF
    assert int(synth.splitmix64(np.uint64(1))) == 0x910A2DEC89025CC1


def test_cost_map_is_deterministic_and_in_range():
    a = synth.cost_map(7, 256, 192)
    b = synth.cost_map(7, 256, 192)
    assert a.shape == (192, 256) and a.dtype == np.uint8
    assert np.array_equal(a, b)
    free = a[a != 255]
    assert free.min() >= 1 and free.max() <= 200
    assert 0.002 < (a == 255).mean() < 0.05
    (sx, sy), (gx, gy) = synth.start_goal(256, 192)
    assert (a[int(sx) - 16:int(sx) + 17, int(sy) - 16:int(sy) + 17] != 255).all()
    assert (a[int(gx) - 16:int(gx) + 17, int(gy) - 16:int(gy) + 17] != 255).all()
    assert hashlib.md5(synth.cost_map(7, 256, 256).tobytes()).hexdigest() == "d977490d7d9864eae5425112d5d28637"


def test_replan_script_shapes():
    s = list(synth.replan_script(7, 300, 200, n_patches=100))
    assert len(s) == 100
    for k, (px, py), top, left, patch in s:
        assert patch.shape == (31, 31) and patch.min() >= 1 and patch.max() <= 200
        assert 0 <= top <= 200 - 31 and 0 <= left <= 300 - 31
        assert 0 <= px < 200 and 0 <= py < 300


def test_maps_are_c_contiguous():
    """consumers hand out the raw buffer (bench.py: torch.from_numpy(cost).to(device).data_ptr());
    a column-major array there is the transposed map"""
    import ufm_amd
    c = ufm_amd.synth.cost_map(7, 96, 64)
    assert c.flags["C_CONTIGUOUS"] and c.shape == (64, 96)
    for k, s, top, left, patch in ufm_amd.synth.replan_script(7, 96, 64, n_patches=2):
        assert patch.flags["C_CONTIGUOUS"]
