"""CPU: BASELINE.json configs[0] as specified -- `rife -t 2` on a 16-frame 480p (854x480) synthetic clip, scene detection
off, the whole driver loop (drba_amd.infer.interpolate_stream, reference infer.py:58-174) on the fp32 CPU path.

The product has no CPU compute path (by design: ops raise on CPU tensors), so the model here is the oracle -- the CPU
restatement pinned bit-for-bit to the reference -- behind the product's driver, schedule, size logic and frame
conversion semantics.  Checked: 32 frames written, of which 28 come from 14 inference_ts_drba steps, 1 + 1 from
inference_ts (head / tail) and 2 are pass-through copies of the first and last source frame;
net size 512x896; the written frames are uint8 854x480 and the DRBA frames lie between their source neighbours."""
import numpy as np
import torch

from drba_amd import infer as drv
from drba_amd.models.utils import tools
from drba_amd.utils import synth
from tests.clip_common import ListIO, CountingModel, cpu_hooks


def test_config1_rife_t2_480p_16_frames_cpu():
    import oracle
    torch.set_num_threads(8)
    frames = synth.make_clip(16, 480, 854, seed=1234)
    size = tools.get_valid_net_inp_size(frames[0], 1.0, div=64)
    assert size == {"src_size": (480, 854), "dst_size": (512, 896)}
    model = CountingModel(oracle.rife.RifeOracle(synth.ifnet_state_dict(seed=0), 1.0))
    io = ListIO(frames, 24.0)
    to_inp, to_out, check = cpu_hooks()
    n = drv.interpolate_stream(model, io, 48.0, times=2, enable_scdet=False, to_inp=to_inp, to_out=to_out, check_scene=check)
    assert n == 32 and len(io.written) == 32
    assert model.calls == {"inference_ts": 2, "inference_ts_drba": 14}
    assert model.generated == {"inference_ts": 2, "inference_ts_drba": 28}
    assert all(f.shape == (480, 854, 3) and f.dtype == np.uint8 for f in io.written)
    # -t 2: head = [copy of frame 0, frame at t=0.25 of (0,1)], tail = [t=0.75 of (14,15), copy of frame 15]
    assert np.array_equal(io.written[0], to_out(to_inp(frames[0], (512, 896)), (480, 854)))
    assert np.array_equal(io.written[-1], to_out(to_inp(frames[15], (512, 896)), (480, 854)))
    # a synthesised frame is closer to its centre source frame than the two neighbouring source frames are to each other
    k = 7  # outputs 2k, 2k+1 straddle source frame k (t = 0.75 / 1.25 around it)
    a, b, c = [f.astype(np.float32) for f in (frames[k - 1], frames[k], frames[k + 1])]
    for out in (io.written[2 * k], io.written[2 * k + 1]):
        assert np.abs(out.astype(np.float32) - b).mean() < max(np.abs(a - b).mean(), np.abs(c - b).mean())
