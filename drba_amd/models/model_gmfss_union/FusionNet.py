"""GridNet on the HIP library (reference models/model_gmfss_union/FusionNet.py:6-146, same state-dict keys).

Every block is (PReLU, conv | deconv4x4 s2, PReLU, conv3x3); the PReLUs are applied inside the conv loaders and the
grid's lateral / vertical sums are extra operands of the second conv's epilogue, so no activation or add ever
makes its own trip through HBM."""
from drba_amd import ops as _ops


class _TwoConv:
    def __init__(self, sd, p, device, stride1=1, transposed=False):
        if transposed:
            self.first = _ops.Deconv4x4(sd[p + "1.weight"], sd[p + "1.bias"], pixel_shuffle=False, device=device,
                                        pre_slope=float(sd[p + "0.weight"]))
        else:
            self.first = _ops.Conv3x3(sd[p + "1.weight"], sd[p + "1.bias"], stride=stride1, act=None, device=device,
                                      pre_slope=float(sd[p + "0.weight"]))
        self.second = _ops.Conv3x3(sd[p + "3.weight"], sd[p + "3.bias"], stride=1, act=None, device=device,
                                   pre_slope=float(sd[p + "2.weight"]))

    def __call__(self, x, add=None, add2=None):
        """block(x) [+ add [+ add2]] with the sums formed in the epilogue in that order."""
        return self.second(self.first(x), residual=add, residual2=add2)


class GridNet:
    def __init__(self, sd, device):
        R = lambda n: _TwoConv(sd, f"residual_model_{n}.", device)  # noqa: E731
        D = lambda n: _TwoConv(sd, f"downsample_model_{n}.", device, stride1=2)  # noqa: E731
        U = lambda n: _TwoConv(sd, f"upsample_model_{n}.", device, transposed=True)  # noqa: E731
        head0 = "head0" if "residual_model_head0.0.weight" in sd else "head"  # model_gmfss names it "head"
        self.head = [R(head0), R("head1"), R("head2"), R("head3")]
        self.r = {n: R(n) for n in ("01", "04", "05", "11", "14", "15", "21", "24", "25")}
        self.d = {n: D(n) for n in ("10", "20", "11", "21")}
        self.u = {n: U(n) for n in ("04", "14", "05", "15")}
        p = "residual_model_tail."
        self.tail_a = _ops.Conv3x3(sd[p + "conv_before_upsample.0.weight"], sd[p + "conv_before_upsample.0.bias"],
                                   act="prelu", post_slope=float(sd[p + "conv_before_upsample.1.weight"]), device=device)
        self.tail_up = _ops.Conv3x3(sd[p + "upsample.0.weight"], sd[p + "upsample.0.bias"], act=None, device=device)
        self.tail_last = _ops.Conv3x3(sd[p + "conv_last.weight"], sd[p + "conv_last.bias"], act=None, device=device)

    def __call__(self, x, x1, x2, x3):
        r, d, u = self.r, self.d, self.u
        X00 = self.head[0](x, add=self.head[1](x1))
        X01 = r["01"](X00, add=X00)
        X10 = d["10"](X00, add=self.head[2](x2))
        X20 = d["20"](X10, add=self.head[3](x3))
        X11 = r["11"](X10, add=X10, add2=d["11"](X01))
        X21 = r["21"](X20, add=X20, add2=d["21"](X11))
        X24 = r["24"](X21, add=X21)
        X25 = r["25"](X24, add=X24)
        X14 = r["14"](X11, add=X11, add2=u["14"](X24))
        X04 = r["04"](X01, add=X01, add2=u["04"](X14))
        X15 = r["15"](X14, add=X14, add2=u["15"](X25))
        X05 = r["05"](X04, add=X04, add2=u["05"](X15))
        # upsample conv + nn.PixelShuffle(2) (FusionNet.py:100-103): the shuffle is the convolution's store where a tile has that form
        return self.tail_last(_ops.conv3x3_shuffle(self.tail_up, self.tail_a(X05)))

    forward = __call__
