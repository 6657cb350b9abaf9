"""GPU: toc3d_linear_chain (attn.proj -> w1|w2 -> w3 of one block half in ONE persistent launch, eva_vit.py:44-51,115,262-263) is bit-identical
to the same ops issued as separate toc3d_linear_fused launches -- for every chain config, ragged sizes, many bands, repeated launches on one
state buffer, and while another stream keeps the chip unevenly busy (hand-offs are tested under load, cdna_hip_programming.md Guideline 16)."""
import os

import pytest
import torch

from toc3d_amd import lib

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not lib.experimental(), reason="round-3 experiment: `make EXPERIMENTAL=1`, TOC3D_LIB=libtoc3d_gfx950_exp.so")]
DEV = "cuda:0"


def S():
    return torch.cuda.current_stream().cuda_stream


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def ru(a, b):
    return (a + b - 1) // b * b


class Block:
    """Packed weights + buffers of one block half at (M, C, Hd)."""

    def __init__(self, M, C, Hd, seed=0):
        dt, tdt = lib.BF16, torch.bfloat16
        self.M, self.C, self.Hd = M, C, Hd
        self.Hp = Hp = ru(Hd, 64)
        self.eps = 1e-6
        d = lambda t: t.to(DEV)
        self.att = d(rnd(M, C, seed=seed + 1)).to(tdt)
        wp = torch.zeros(ru(C, 128), C, dtype=tdt, device=DEV)
        lib.call("toc3d_pack_weight", dt, d(rnd(C, C, seed=seed + 2, scale=C ** -0.5)), C, C, wp, wp.shape[0], C, S())
        self.wproj, self.bproj = wp, d(rnd(C, seed=seed + 3))
        self.x0 = d(3.0 * rnd(M, C, seed=seed + 4) + 0.7)
        g2, b2 = d(1.0 + 0.3 * rnd(C, seed=seed + 5)), d(0.2 * rnd(C, seed=seed + 6))
        w1, w2 = d(rnd(Hd, C, seed=seed + 7, scale=C ** -0.5)), d(rnd(Hd, C, seed=seed + 8, scale=C ** -0.5))
        bb1, bb2 = d(rnd(Hd, seed=seed + 9)), d(rnd(Hd, seed=seed + 10))
        self.w12f = torch.empty(2 * Hp, C, dtype=tdt, device=DEV)
        self.c1_12, self.c2_12 = torch.empty(2 * Hp, device=DEV), torch.empty(2 * Hp, device=DEV)
        lib.call("toc3d_pack_swiglu_lnfold", dt, w1, w2, bb1, bb2, g2, b2, Hd, C, self.w12f, self.c1_12, self.c2_12, Hp, C, S())
        self.w12 = torch.empty(2 * Hp, C, dtype=tdt, device=DEV)
        self.b12 = torch.empty(2 * Hp, device=DEV)
        lib.call("toc3d_pack_swiglu", dt, w1, w2, bb1, bb2, Hd, C, self.w12, self.b12, Hp, C, S())
        gf, bf = d(1.0 + 0.3 * rnd(Hd, seed=seed + 11)), d(0.2 * rnd(Hd, seed=seed + 12))
        W3, b3 = d(rnd(C, Hd, seed=seed + 13, scale=Hd ** -0.5)), d(rnd(C, seed=seed + 14))
        self.w3f = torch.zeros(ru(C, 128), Hp, dtype=tdt, device=DEV)
        self.c1, self.c2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        lib.call("toc3d_pack_weight_lnfold", dt, W3.contiguous(), gf, bf, b3, C, Hd, self.w3f, self.w3f.shape[0], Hp, self.c1, self.c2, S())
        self.rep_index = torch.full((M,), -1, dtype=torch.int32, device=DEV)
        self.rep_index[::37] = torch.arange(len(range(0, M, 37)), dtype=torch.int32, device=DEV)
        self.nrep = int((self.rep_index >= 0).sum())
        self.cap2, self.cap = C // 64, -(-2 * Hp // 128)
        self.a_ln = d(rnd(M, C, seed=seed + 15)).to(tdt)                # family 1: the rows a LayerNorm launch left
        self.fresh()

    def fresh(self):
        M, C, Hp = self.M, self.C, self.Hp
        tdt = torch.bfloat16
        self.x = self.x0.clone()
        self.a = torch.full((M, C), 9.0, dtype=tdt, device=DEV)
        self.hid = torch.full((M, Hp), 9.0, dtype=tdt, device=DEV)
        self.st2 = torch.zeros(4 + M * self.cap2 * 2, device=DEV)
        self.st = torch.zeros(4 + M * self.cap * 2, device=DEV)
        self.rep1 = torch.zeros(self.nrep, C, device=DEV)
        self.rep2 = torch.zeros(self.nrep, C, device=DEV)

    def ops(self, family):
        M, C, Hd, Hp = self.M, self.C, self.Hd, self.Hp
        proj = (lib.EPI_RESIDUAL_STATS, self.att, C, self.wproj, C, self.bproj, self.x, C, self.x, C, self.rep1, self.rep_index, M, C, C, 0,
                (self.st2, self.cap2, None, 0, None, 0, 0.0, self.a, C, None))
        w12 = (lib.EPI_SWIGLU_STATS_LN, self.a, C, self.w12f, C, self.c2_12, self.hid, Hp, None, 0, None, None, M, 2 * Hp, C, Hd,
               (self.st, self.cap, self.st2, self.cap2 | self.cap2 << 32, self.c1_12, C, self.eps, None, 0, None))
        w12_plain = (lib.EPI_SWIGLU_STATS, self.a_ln, C, self.w12, C, self.b12, self.hid, Hp, None, 0, None, None, M, 2 * Hp, C, Hd,
                     (self.st, self.cap, None, 0, None, 0, 0.0, None, 0, None))
        w3 = (lib.EPI_RESIDUAL_LN, self.hid, Hp, self.w3f, Hp, self.c2, self.x, C, self.x, C, self.rep2, self.rep_index, M, C, Hp, 0,
              (None, 0, self.st, self.cap | self.cap << 32, self.c1, Hd, self.eps, None, 0, None))
        return [proj, w12, w3] if family == 0 else [w12_plain, w3]

    def run_separate(self, family, variant=16):
        for o in self.ops(family):
            epi, rest, fused = o[0], o[1:16], o[16]
            A, lda, W, ldw, bias, out, ldo, res, ldr, rep_out, rep_index, M, N, K, nv = rest
            lib.call("toc3d_linear_fused", lib.BF16, epi, variant, A, lda, W, ldw, bias, out, ldo, res, ldr, 0, rep_out, rep_index, M, N, K, nv, *fused, S())

    def prepare_chain(self, config, state, n_bands=8, grid=768, flags=0, **sched_kw):
        """Everything host-side done once (schedule upload, argument blocks); returns the launch closure."""
        family = config // 10
        ops = self.ops(family)
        Ns = [o[13] for o in ops]
        sched, nb = lib.chain_schedule(config, self.M, Ns, n_bands=n_bands, **sched_kw)
        sched_t = torch.tensor(sched, dtype=torch.int32, device=DEV)
        keep = (sched_t, state)

        def launch():
            cops = [lib.chain_op(o[0], *o[1:16], fused=o[16]) for o in self.ops(family)]     # buffers are re-made by fresh()
            lib.linear_chain(lib.BF16, config, cops, keep[0], nb, state, grid, flags, S())
        return launch

    def run_chain(self, config, state, **kw):
        self.prepare_chain(config, state, **kw)()

    def snapshot(self):
        return [t.clone() for t in (self.x, self.a, self.hid, self.st2[4:], self.st[4:], self.rep1, self.rep2)]


def new_state():
    return torch.zeros(lib.CHAIN_STATE_BYTES // 4, dtype=torch.int32, device=DEV)


def check_equal(got, ref, what):
    names = ("x", "a", "hid", "stats2", "stats", "rep1", "rep2")
    for n, g, r in zip(names, got, ref):
        assert torch.equal(g, r), f"{what}: {n} differs from the separate launches ({(g.float() - r.float()).abs().max().item():.3e})"


@pytest.mark.parametrize("M,C,Hd", [(777, 384, 300), (130, 256, 200), (1500, 512, 700)])
@pytest.mark.parametrize("config", [0, 1, 2, 10, 11, 12])
def test_chain_equals_separate_launches(config, M, C, Hd):
    b = Block(M, C, Hd)
    family = config // 10
    b.run_separate(family)
    ref = b.snapshot()
    state = new_state()
    for n_bands, grid, flags, kw in ((8, 768, 0, {}), (3, 64, 1, {}), (64, 512, 0, dict(lag=2)), (1, 16, 0, {}), (8, 768, 0, dict(n_major=(1,) if family == 0 else (0,)))):
        b.fresh()
        b.run_chain(config, state, n_bands=n_bands, grid=grid, flags=flags, **kw)
        torch.cuda.synchronize()
        assert lib.chain_status(state) == 0, f"chain error code {lib.chain_status(state)}"       # the sticky word callers poll at their sync points
        assert int(state.abs().sum().item()) == 0, "the last workgroup re-arms the state"
        check_equal(b.snapshot(), ref, f"config {config} bands {n_bands} grid {grid} flags {flags} {kw}")


def test_chain_at_vitl_size_repeated_and_under_uneven_load():
    """ViT-L block half at a frame's accelerated-block size: 30 launches on one state buffer while a second stream runs bursts of other work
    (so that workgroups arrive unevenly and consumers are L1-warm); every launch bit-identical to the separate launches."""
    M, C, Hd = 2898, 1024, 2730
    b = Block(M, C, Hd)
    b.run_separate(0)
    ref = b.snapshot()
    state = new_state()
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=DEV)
    for it in range(30):
        b.fresh()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(it % 4):
                junk = torch.tanh(junk @ junk[:, :4096] * 1e-3)
        b.run_chain(0 if it % 3 else 1, state, n_bands=8 + (it % 5) * 4, grid=512 + 64 * (it % 5), flags=it % 2)
        torch.cuda.synchronize()
        assert int(state[1].item()) == 0
        check_equal(b.snapshot(), ref, f"launch {it}")


def test_chain_timing_report():
    """Not an assertion of speed -- prints the chain against the separate launches at the frame's sizes (cold operands between repetitions)."""
    flush = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=DEV)
    for M in (2898, 3744, 6000):
        b = Block(M, 1024, 2730)
        state = new_state()

        def timed(fn, reps=7):
            ts = []
            for _ in range(reps):
                b.fresh()
                flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            return min(ts), sorted(ts)[len(ts) // 2]
        sep = timed(lambda: b.run_separate(0))
        print(f"[chain M={M}] separate launches (variant 16): min {sep[0]:.1f} us, median {sep[1]:.1f} us")
        for config in (0, 1, 2):
            for nb, grid, kw in ((8, 768, {}), (16, 768, {}), (8, 512, {}), (8, 768, dict(lag=2)), (8, 768, dict(n_major=(1,)))):
                if config == 2 and grid > 512:
                    grid = 512
                t = timed(b.prepare_chain(config, state, n_bands=nb, grid=grid, **kw))
                print(f"[chain M={M}] config {config} bands {nb} grid {grid} {kw}: min {t[0]:.1f} us, median {t[1]:.1f} us")
        assert int(state[1].item()) == 0
