"""Edge-case sweeps GPU vs CPU oracle (free-running up to the first window closure, where nothing depends on CCIPCA's
irreproducible second component): tiny dimensions with every J (ncol = J + eigen_method_one may exceed D), the flag
matrix, all four sampler variants; and funnel dimensions around the lane/octet boundary and the kernels' chunk sizes."""
import itertools

import numpy as np
import pytest

from parity_util import make_sampler

pytestmark = pytest.mark.gpu


def test_tiny_dimensions_every_J_and_flag(oracle):
    O = oracle
    for D, J, emo, sdc, variant in itertools.product((1, 2, 3, 5), (1, 2, 3), (0, 1), (0, 1), (0, 1, 2, 3)):
        kw = dict(warmup=100, J=J, eigen_method_one=bool(emo), scale_dir_cov=bool(sdc))
        if variant == 3:
            c = O.SliceChain("normal", {"D": D}, seed=11, **kw)
            s = make_sampler("normal", {"D": D}, 2, variant=3, tol=1e-12, **kw)
        else:
            tol, sc = (1e-12, 600) if variant == 1 else (1e-10, 300)          # klhr.py's own defaults for the Gaussian variant
            c = O.Chain("normal", {"D": D}, seed=11, variant={0: "sinh", 1: "gauss", 2: "sub"}[variant], tol=tol, scale_clip=sc, **kw)
            s = make_sampler("normal", {"D": D}, 2, variant=variant, tol=tol, scale_clip=sc, **kw)
        s.seed(np.array([11, 12], dtype=np.uint64))
        s.step(49, record_draws=True)
        dg = s.draws(0, 49)[:, 0, :]
        do = np.array([c.draw() for _ in range(49)])
        tag = (D, J, emo, sdc, variant)
        assert np.abs(dg - do).max() < 1e-7, tag
        s.step(2)
        [c.draw() for _ in range(2)]                                           # across the first closure (draw 50)
        ev_g, ev_o = s.get("eigvals")[0], c.get("eigvals")
        assert np.allclose(s.get("cov")[0], c.get("cov"), rtol=1e-6, atol=1e-9), tag
        assert np.allclose(ev_g[0], ev_o[0], rtol=1e-5), tag
        if emo:
            assert ev_g[-1] == 1.0 and ev_o[-1] == 1.0, tag                    # the extra slot is never overwritten
        assert not s.get("status").any(), tag


def test_funnel_dimensions_around_kernel_boundaries(oracle):
    for Da in (1, 2, 6, 7, 8, 9, 12, 13, 14, 15, 16, 17, 24, 25, 26, 27, 31, 32, 33, 63, 64, 65, 103, 104, 105):
        c = oracle.Chain("funnel", {"D": Da}, seed=21, warmup=100)
        s = make_sampler("funnel", {"D": Da}, 3, warmup=100)
        s.seed(np.array([21, 22, 23], dtype=np.uint64))
        s.step(49, record_draws=True)
        dg = s.draws(0, 49)[:, 0, :]
        do = np.array([c.draw() for _ in range(49)])
        # (the 2-D funnel amplifies rounding differences fastest: 4e-6 after 49 draws)
        assert np.abs(dg - do).max() < (1e-4 if Da == 1 else 1e-6), (Da, s.kernel_name())
        s.step(3)
        [c.draw() for _ in range(3)]
        assert np.allclose(s.get("cov")[0], c.get("cov"), rtol=1e-5, atol=1e-9), Da
        assert not s.get("status").any(), Da


def test_window_schedules(oracle):
    """WindowedAdaptation corner cases (no window at all, warmup == windowsize, windowscale 1 with 80 windows, closures at
    the last warm-up draw): after 230 draws the bookkeeping (draw, OnlineMoments N, CCIPCA n, window index) equals the oracle's."""
    for warmup, ws, wsc in itertools.product((0, 10, 49, 50, 51, 100, 149, 150, 151, 400), (5, 50, 60), (1, 2, 3)):
        kw = dict(warmup=warmup, windowsize=ws, windowscale=wsc)
        c = oracle.Chain("funnel", {"D": 9}, seed=5, **kw)
        s = make_sampler("funnel", {"D": 9}, 2, **kw)
        s.seed(np.array([5, 6], dtype=np.uint64))
        s.step(230)
        c.draw_many(230)
        assert list(s.get("counters")[0]) == [230, int(c.get("om_N")), int(c.get("pca_n")), int(c.get("cidx"))], (warmup, ws, wsc)
        assert np.isfinite(s.get("theta")).all()
