"""The synthetic scene generator: determinism (portable counter-based PRNG), sizes, geometry."""
import numpy as np

from lifcal_amd import scene
from tests.helpers import S


def test_prng_is_a_pure_function_of_seed_stream_counter():
    a = scene.Stream(7, 3); b = scene.Stream(7, 3)
    x = a.uniform(10); y = np.concatenate([b.uniform(4), b.uniform(6)])
    assert np.array_equal(x, y)
    assert not np.array_equal(scene.Stream(7, 4).uniform(10), x)
    # known-answer values pin the generator itself (splitmix64 finaliser)
    assert scene._splitmix64(np.array([0], np.uint64))[0] == np.uint64(0xE220A8397B1DCDAF)
    n = scene.Stream(1, 1).normal(200_000)
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1.0) < 0.01


def test_scene_is_reproducible_and_sized():
    s1 = scene.make_scene(S(6, 40, None, 0x506, 701)); s2 = scene.make_scene(S(6, 40, None, 0x506, 701))
    for f in ("u", "v", "mcx", "mcy", "pt", "fr", "cam0", "views0", "pts0"):
        assert np.array_equal(getattr(s1, f), getattr(s2, f)), f
    assert 4.0 < s1.n_obs / (6 * 40) < 8.0                         # ~6 micro images per (point, frame)
    assert s1.pt.max() < 40 and s1.fr.max() < 6
    assert np.all(np.diff(s1.fr.astype(int)) >= 0)                 # reference order: frame-major
    assert np.all(s1.mcx == s1.mcx.astype(np.float32))             # lens centres are float-valued doubles


def test_observations_lie_inside_their_micro_image():
    sc = scene.make_scene(S(6, 40, None, 0xF06, 702))
    d2 = (sc.u - sc.mcx) ** 2 + (sc.v - sc.mcy) ** 2
    assert np.all(d2 < (sc.spec.lens_diameter / 2) ** 2)           # validity radius + noise
    assert np.all((sc.u > -1) & (sc.u < sc.spec.raw_width) & (sc.v > -1) & (sc.v < sc.spec.raw_height))


def test_lens_grid_is_hexagonal():
    g = scene.make_lens_grid(scene.SceneSpec(1, 1)).astype(np.float64)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(g).query(g, k=7)
    inner = (g[:, 0] > 200) & (g[:, 0] < 1800) & (g[:, 1] > 200) & (g[:, 1] < 1800)
    assert np.allclose(d[inner, 1:], 23.2, atol=0.02)              # six neighbours at one lens pitch


def test_scene_with_the_reference_generators_lens_selection():
    """make_scene(lens_selector=...): the lenses that see a point come from the reference's own walk (nearest lens + epipolar web,
    src/CameraCalibration.cpp:661-752, here the oracle's restatement) instead of the K-nearest stand-in; observation values stay the
    forward model at ground truth + noise.  The web finds a subset of what the stand-in accepts (~4 % fewer)."""
    from tests.helpers import S, oracle_lens_selector
    spec = S(6, 60, None, 0xF06, 4711)
    a = scene.make_scene(spec)
    b = scene.make_scene(spec, lens_selector=oracle_lens_selector(spec))
    assert 0.85 * a.n_obs < b.n_obs <= 1.02 * a.n_obs
    key = lambda sc: set(zip(sc.pt.tolist(), sc.fr.tolist(), np.round(sc.mcx, 3).tolist(), np.round(sc.mcy, 3).tolist()))
    ka, kb = key(a), key(b)
    assert len(kb - ka) <= 0.03 * len(kb)          # (doubled web lines of a rotated grid list a few lenses twice; the set view drops them)
    assert np.array_equal(a.cam0, b.cam0) and np.array_equal(a.pts_gt, b.pts_gt) and np.array_equal(a.img_x, b.img_x)
    # every observation lies inside its micro image and is the model's projection up to the noise
    d = np.hypot(b.u - b.mcx, b.v - b.mcy)
    assert d.max() < spec.lens_diameter / 2 - 1.0 + 5 * spec.noise_px + 1e-9
