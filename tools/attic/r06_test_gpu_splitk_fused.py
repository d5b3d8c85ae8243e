"""Split-K finished by the last-arriving block of a tile (AED_OP_CONV_GEMM flag bit 9, csrc/conv_gemm_x6.hip; round 6).

A split-K contraction used to be two launches: blockIdx.z writes its partial tile into a slab of the workspace, a reduce launch sums
the slabs in z order and runs the epilogue.  With a zeroed counter header in front of the workspace the last block to arrive at a tile
does both -- 91 of the 582 launches of an edit-lane step disappear.  Same summation order, so the two forms must agree BIT FOR BIT;
the header must survive records that do not use it (fp32 split-K on the same tape) and reset itself; and because the z blocks of a
tile sit on different XCDs (release / acquire at agent scope around the counter) the launches are repeated under the co-residency
stressor of tests/test_gpu_coresidency.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from audioeditingcode_amd import _lib as L, tape as tape_mod                  # noqa: E402
from audioeditingcode_amd.streams import PartitionStream                      # noqa: E402
from audioeditingcode_amd.tape import Tape                                    # noqa: E402

DEV = torch.device("cuda:0")


def _operands(B, H, W, Cin, N, k, stride, seed, res, rowvec):
    g = torch.Generator().manual_seed(seed)
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = (torch.randn(B, H, W, Cin, generator=g) * torch.exp(0.5 * torch.randn(Cin, generator=g))).to(DEV)
    w = (torch.randn(N, k * k * Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV)
    r = torch.randn(B, OH, OW, N, generator=g).to(DEV) if res else None
    rv = torch.randn(B, N, generator=g).to(DEV) if rowvec else None
    return x, w, b, r, rv, OH, OW, pad


def _record(tp, out, x, w, b, r, rv, *, B, H, W, Cin, N, k, stride, OH, OW, pad, tile, ks, act):
    tp.conv(x, w, b, out, B=B, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW, N=N, KH=k, KW=k, stride=stride, pad_h=pad, pad_w=pad, res=r,
            rowvec=rv, ld_rv=N if rv is not None else 0, out_act=act, tile=tile, ksplit=ks)


# (B, H, W, Cin, N, k, stride, tile, ksplit, residual, per-batch row vector, activation): the edit lane's split shapes (U-Net levels
# 3 / 2 / 1 at batch 2: tiles 3 = 64x128 and 4 = 64x64), ragged M / N, the 128x128 tile, a split deeper than the chunk count allows
CASES = [(2, 32, 2, 640, 640, 3, 1, 3, 16, True, False, 0), (2, 64, 4, 384, 384, 3, 1, 3, 8, False, True, 0),
         (2, 128, 8, 256, 256, 3, 1, 3, 4, True, False, 0), (1, 128, 1, 3200, 640, 1, 1, 4, 8, True, False, 0),
         (2, 64, 4, 384, 384, 3, 2, 4, 8, False, False, L.ACT_SILU), (3, 10, 7, 96, 200, 3, 1, 4, 5, True, True, 0),
         (2, 32, 8, 256, 256, 3, 1, 1, 6, True, False, 0), (1, 64, 1, 64, 64, 1, 1, 4, 32, False, False, 0)]


@pytest.mark.parametrize("B,H,W,Cin,N,k,stride,tile,ks,res,rowvec,act", CASES)
def test_fused_split_k_is_bit_identical_to_the_reduce_launch(B, H, W, Cin, N, k, stride, tile, ks, res, rowvec, act):
    x, w, b, r, rv, OH, OW, pad = _operands(B, H, W, Cin, N, k, stride, 100 * tile + ks, res, rowvec)
    geo = dict(B=B, H=H, W=W, Cin=Cin, N=N, k=k, stride=stride, OH=OH, OW=OW, pad=pad, tile=tile, ks=ks, act=act)
    outs = {}
    for fuse in (0, 1):
        tape_mod.FUSE_SPLITK = fuse
        try:
            tp = Tape(DEV)
            out = tp.alloc(3, B, OH, OW, N, zero=True)
            with tape_mod.arith_mode("bf16x6"):
                for rep in range(3):                           # three launches through one workspace: the header resets itself
                    _record(tp, out[rep], x, w, b, r, rv, **geo)
        finally:
            tape_mod.FUSE_SPLITK = 1
        assert all(bool(op.flags & 512) == bool(fuse) and op.flags & 4 and op.i[28] == ks for op in tp.ops)
        tp.run()
        torch.cuda.synchronize()
        if fuse:
            assert int(tp.ws[:tape_mod.SPLITK_HEADER].view(torch.int32).abs().sum()) == 0, "arrival counters did not reset"
        outs[fuse] = out.clone()
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[1][0]) and torch.equal(outs[1][2], outs[1][0])
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.view(N, k, k, Cin).permute(0, 3, 1, 2).double(), b.double(), stride=stride,
                   padding=pad).permute(0, 2, 3, 1)
    if rv is not None:
        ref = ref + rv.double()[:, None, None, :]
    if r is not None:
        ref = ref + r.double()
    if act == L.ACT_SILU:
        ref = F.silu(ref)
    rel = float((outs[1][0].double() - ref).norm() / ref.norm())
    assert rel < 2e-6, rel


def test_counter_header_survives_records_that_do_not_use_it():
    """One tape, one workspace: fused split-bf16 record, an fp32 split-K record (reduce launch, slabs behind the header), a fused
    record of another shape -- in a hipGraph, replayed: every replay reproduces the eager results."""
    a = _operands(2, 32, 2, 640, 640, 3, 1, 1, True, False)
    c = _operands(2, 64, 4, 384, 384, 1, 1, 2, False, False)
    tp = Tape(DEV)
    o1, o2, o3 = tp.alloc(2, 32, 2, 640), tp.alloc(2, 32, 2, 640), tp.alloc(2, 64, 4, 384)
    ga = dict(B=2, H=32, W=2, Cin=640, N=640, k=3, stride=1, OH=a[5], OW=a[6], pad=a[7], act=0)
    gc = dict(B=2, H=64, W=4, Cin=384, N=384, k=1, stride=1, OH=c[5], OW=c[6], pad=c[7], act=0)
    with tape_mod.arith_mode("bf16x6"):
        _record(tp, o1, *a[:5], tile=3, ks=16, **ga)
    with tape_mod.arith_mode("f32"):
        _record(tp, o2, *a[:5], tile=4, ks=8, **ga)
    with tape_mod.arith_mode("bf16x6"):
        _record(tp, o3, *c[:5], tile=4, ks=4, **gc)
    assert [bool(op.flags & 512) for op in tp.ops] == [True, False, True]
    st = torch.cuda.Stream(DEV)
    with torch.cuda.stream(st):
        tp.run()
        st.synchronize()
        first = [t.clone() for t in (o1, o2, o3)]
        tp.capture()
        for _ in range(5):
            for t in (o1, o2, o3):
                t.zero_()
            tp.replay()
            st.synchronize()
            assert all(torch.equal(t, f) for t, f in zip((o1, o2, o3), first))
    assert float((o1 - o2).norm() / o2.norm()) < 3e-6          # the two arithmetics agree on the same convolution


@pytest.mark.parametrize("case", [0, 1, 3])
def test_fused_split_k_launches_are_identical_under_a_co_resident_stressor(case):
    """R = 200 launches on a 64-CU masked stream (an edit lane) while split-bf16 convolutions of an unmasked stream share the CUs:
    every output equals the solo launch bit for bit."""
    from test_gpu_coresidency import _stressor
    B, H, W, Cin, N, k, stride, tile, ks, res, rowvec, act = CASES[case]
    R = 200
    x, w, b, r, rv, OH, OW, pad = _operands(B, H, W, Cin, N, k, stride, 7 + case, res, rowvec)
    gen = torch.Generator(device=DEV)
    gen.manual_seed(case)
    stress = _stressor(gen)
    lane = PartitionStream.acquire(DEV, cus=range(0, 64), total=256, index=0)
    side = PartitionStream.acquire(DEV, index=17)
    outs = torch.zeros(R + 1, B, OH, OW, N, device=DEV)
    tp = Tape(DEV)
    with tape_mod.arith_mode("bf16x6"):
        for q in range(R + 1):
            _record(tp, outs[q], x, w, b, r, rv, B=B, H=H, W=W, Cin=Cin, N=N, k=k, stride=stride, OH=OH, OW=OW, pad=pad, tile=tile,
                    ks=ks, act=act)
    tp.finalize()
    assert all(op.flags & 512 for op in tp.ops)
    with torch.cuda.stream(lane.stream):
        tp.run(0, 1)
    torch.cuda.synchronize()
    with torch.cuda.stream(side.stream):
        for _ in range(R // 8):
            stress.run()
    with torch.cuda.stream(lane.stream):
        tp.run(1, R + 1)
    torch.cuda.synchronize()
    bad = [q for q in range(1, R + 1) if not torch.equal(outs[q], outs[0])]
    assert not bad, (len(bad), R, float((outs[bad[0]] - outs[0]).abs().max()))
