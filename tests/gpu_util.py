"""Helpers for the -m gpu parity tests: pitched device surfaces backed by torch uint8 tensors.

torch is used only as a device allocator / copy engine here; every conversion goes through the C ABI.
"""
from __future__ import annotations

import numpy as np
import torch

PAD = 0xCD


def round_up(v, a):
    return (v + a - 1) // a * a


class DevPlanes:
    """Device copy of a list of numpy planes, each in its own pitched allocation.

    pitch = round_up(row_bytes, align) + extra; padding bytes are 0xCD and can be verified untouched.
    `offset` shifts the first byte of every plane (to exercise unaligned base pointers).
    """

    def __init__(self, host_planes, align=256, extra=0, offset=0, device="cuda:0"):
        self.host_shapes = [(p.shape, p.dtype) for p in host_planes]
        self.bufs, self.pitches, self.offset = [], [], offset
        for p in host_planes:
            rows, rb = p.shape[0], p.shape[1] * p.dtype.itemsize
            pitch = round_up(rb, align) + extra
            h = np.full((rows * pitch + offset + 64,), PAD, dtype=np.uint8)
            view = h[offset:offset + rows * pitch].reshape(rows, pitch)
            view[:, :rb] = p.view(np.uint8).reshape(rows, rb)
            self.bufs.append(torch.from_numpy(h).to(device))
            self.pitches.append(pitch)

    def desc(self):
        return [(b.data_ptr() + self.offset, pitch) for b, pitch in zip(self.bufs, self.pitches)]

    def upload(self, host_planes):
        """new pixels into the SAME device buffers (a captured graph keeps their addresses)"""
        for b, pitch, p in zip(self.bufs, self.pitches, host_planes):
            rows, rb = p.shape[0], p.shape[1] * p.dtype.itemsize
            h = np.full((rows * pitch + self.offset + 64,), PAD, dtype=np.uint8)
            h[self.offset:self.offset + rows * pitch].reshape(rows, pitch)[:, :rb] = p.view(np.uint8).reshape(rows, rb)
            b.copy_(torch.from_numpy(h))

    def download(self):
        """-> (list of tight numpy planes, padding_intact: bool)"""
        out, intact = [], True
        for b, pitch, (shape, dt) in zip(self.bufs, self.pitches, self.host_shapes):
            rows, rb = shape[0], shape[1] * np.dtype(dt).itemsize
            h = b.cpu().numpy()
            body = h[self.offset:self.offset + rows * pitch].reshape(rows, pitch)
            out.append(np.ascontiguousarray(body[:, :rb]).view(dt).reshape(shape))
            intact &= bool((body[:, rb:] == PAD).all()) and bool((h[:self.offset] == PAD).all()) and bool(
                (h[self.offset + rows * pitch:] == PAD).all())
        return out, intact


def stream_handle():
    return torch.cuda.current_stream().cuda_stream


def assert_planes_equal(got, want, what=""):
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        if not np.array_equal(g, w):
            d = np.argwhere(g != w)
            first = tuple(d[0])
            raise AssertionError(
                f"{what}: plane {i} differs at {len(d)} of {g.size} elements; first at {first}: got {g[first]} want {w[first]}")
