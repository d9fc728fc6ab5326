"""Lost packets: concealment in the mode of the last packet -- SILK frames continued from the last frame's parameters (pitch
repetition / noise), CELT noise shaped by decaying band energies, both for hybrid.  The reference has none -- it panics on an
empty packet (packet.rs:215-225) -- so there is no reference output: the oracle and the product implement the same stated
behaviour and are compared with each other, plus the properties below."""
import numpy as np
import pytest

import orc
from gpu_util import check_pcm
from test_emu_silk import EmuDecoder, emu  # noqa: F401  (fixture)


def rms(x):
    return float(np.sqrt(np.mean(np.square(x.astype(np.float64))))) if len(x) else 0.0


@pytest.mark.parametrize("vid,npk", [(1, 300), (4, 300), (6, 300), (10, 300)])
def test_lost_packets_are_concealed(emu, vid, npk):  # noqa: F811
    _, pk = orc.load_vectors(ids=[vid], max_packets=npk)[vid]
    d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
    last, ratios = None, []
    for i, p in enumerate(pk):
        lost = i % 37 == 20 or i % 37 == 21            # single and double losses
        want = ref.decode(b"" if lost else p)
        out = d.decode(b"" if lost else bytes(p))
        assert not isinstance(out, int), (i, out)
        assert out.shape == want.shape
        check_pcm(want, out, f"vector {vid} packet {i}{' (lost)' if lost else ''}")
        if lost:
            assert last is not None and want.shape == last.shape, "a lost packet spans what the last packet spanned"
            assert np.isfinite(want).all() and np.abs(want).max() <= 1.0 + 1e-6
            if i % 37 == 20 and rms(last) > 1e-3:
                ratios.append(rms(want) / rms(last))
        last = want
    # the concealed audio carries on at a level comparable to what came before: neither silence nor a burst
    assert ratios and 0.1 < float(np.median(ratios)) < 2.0, ratios


@pytest.mark.parametrize("vid,start", [(1, 60), (4, 60), (6, 60), (10, 100)])
def test_a_burst_of_losses_fades_out(emu, vid, start):  # noqa: F811
    """twelve packets lost in a row: product == oracle throughout, the level falls from frame to frame, and the first packet
    after the gap decodes from the state the concealment left (no restart)"""
    _, pk = orc.load_vectors(ids=[vid], max_packets=start + 20)[vid]
    d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
    level = []
    for i, p in enumerate(pk):
        lost = start <= i < start + 12
        want, out = ref.decode(b"" if lost else p), d.decode(b"" if lost else bytes(p))
        check_pcm(want, out, f"vector {vid} packet {i}{' (lost)' if lost else ''}")
        if lost:
            level.append(rms(want))
        elif i == start - 1:
            before = rms(want)
    if before > 1e-3:
        assert level[0] > 0.05 * before, (before, level)
        assert level[-1] < 0.25 * max(level[:3]), level


def test_loss_before_any_packet(emu):  # noqa: F811
    d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
    out, want = d.decode(b""), ref.decode(b"")
    assert out.shape == want.shape == (960, 2) and not out.any() and not want.any()


def test_dtx_packet_is_concealed_over_its_own_span(emu):  # noqa: F811
    """a TOC-only packet (one empty frame: DTX, RFC 6716 3.2.1) conceals the duration its TOC names"""
    _, pk = orc.load_vectors(ids=[4], max_packets=40)[4]
    d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
    for p in pk:
        check_pcm(ref.decode(p), d.decode(bytes(p)))
    for toc, n in ((0x08, 960), (0x00, 480), (0x18, 2880), (0xF8, 960), (0xE0, 120)):
        want, out = ref.decode(bytes([toc])), d.decode(bytes([toc]))
        assert want.shape == out.shape == (n, 2)
        check_pcm(want, out, f"TOC {toc:#x}")
