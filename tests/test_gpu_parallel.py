"""GPU: the frame-sharded runner on the HIP model.  gpurun exposes one GPU, so the four ranks of a world of 4 are run one
after the other on cuda:0 (each builds its own halo state exactly as it would on its own GPU) and the concatenation of
their outputs is compared with the sequential HIP driver over the whole clip: same frames, scene cut exactly on a shard
boundary, lookahead active inside the shards, all three model families' warm_reuse."""
import numpy as np
import pytest
import torch

from drba_amd import infer as drv
from drba_amd import parallel
from drba_amd.utils import synth
from tests.clip_common import ListIO

pytestmark = pytest.mark.gpu


def _hooks(dev):
    from drba_amd import ops

    def to_inp(fr, size):
        return ops.to_inp(torch.from_numpy(np.ascontiguousarray(fr)).to(dev), size)

    def to_out(x, size):
        return ops.to_out(x, size).cpu().numpy()

    return to_inp, to_out


def _sharded_vs_sequential(model, frames, dst_fps, times, scdet, world, dev, max_lsb=1, max_frac=1e-4):
    to_inp, to_out = _hooks(dev)
    io = ListIO(frames, 24.0)
    drv.interpolate_stream(model, io, dst_fps, times=times, enable_scdet=scdet, to_inp=to_inp, to_out=to_out)
    counts = parallel.emission_counts(len(frames), 24.0, dst_fps, times, world)
    parts = []
    for rank in range(world):
        mine = parallel.interpolate_shard(model, frames, 24.0, dst_fps, rank, world, times=times, enable_scdet=scdet,
                                          to_inp=to_inp, to_out=to_out)
        assert len(mine) == sum(counts[rank]), (rank, len(mine), counts[rank])  # what StreamedGather sizes its rounds by
        parts += mine
    torch.cuda.synchronize()
    assert len(parts) == len(io.written)
    worst, differing = 0, 0
    for a, b in zip(parts, io.written):
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        worst, differing = max(worst, int(d.max())), differing + int((d > 0).sum())
    total = sum(a.size for a in parts)
    # same kernels on the same inputs; only the atomic arrival order inside the splats differs between the runs
    assert worst <= max_lsb and differing / total < max_frac, (worst, differing / total)


@pytest.mark.parametrize("scdet,times,dst_fps", [(True, -1, 60.0), (False, 2, 48.0)])
def test_rife_world4_sequential_ranks_equal_sequential_driver(hip_backend, scdet, times, dst_fps):
    # 18 frames -> 16 loop iterations -> shards [0,4) [4,8) [8,12) [12,16); the cut between frames 8 and 9 makes
    # iteration 8 (first of rank 2) a cut_right step and iteration 7 (last of rank 1) ... i.e. the boundary state matters
    frames = synth.make_clip(18, 256, 448, seed=31, cut_at=9 if scdet else None)
    model = hip_backend.make_rife(synth.ifnet_state_dict(seed=0), 1.0)
    _sharded_vs_sequential(model, frames, dst_fps, times, scdet, 4, hip_backend.dev)


def test_gmfss_union_world2_sequential_ranks_equal_sequential_driver(hip_backend):
    """GMFSS_UNION's pair state is the 6-tuple of model.reuse; the shard rebuilds it with GMFSS_UNION.warm_reuse."""
    frames = synth.make_clip(7, 128, 256, seed=32)
    model = hip_backend.make_gmfss_union(synth.gmfss_union_state_dicts(seed=0), 1.0)
    # the soft splats' exp(10 tanh) weights and the swap masks are discontinuous decisions: a last-bit difference in a
    # splat sum (atomic arrival order) can move a small patch by up to 5e-2 (gpu_checks.check_gmfss_union) = 13 LSB
    _sharded_vs_sequential(model, frames, 60.0, -1, False, 2, hip_backend.dev, max_lsb=13, max_frac=1e-3)
