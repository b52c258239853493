"""GPU tests of the measurement helpers behind bench.py: synthetic inputs generated on the device must be
the splitmix64 residues the survey specifies, word for word; the VALU clock must be a plausible MI355X clock; the
handle must run on its own device whatever device the caller has current."""
import numpy as np
import pytest

from aloha_amd import vp as avp
from util import splitmix_residues

pytestmark = pytest.mark.gpu

SPM = 0


@pytest.mark.parametrize("logn", [10, 13, 16])
def test_fill_synthetic_is_splitmix64_with_rejection(logn):
    n = 1 << logn
    big = [avp.nth_prime(logn, i) for i in range(3)]
    # moduli far below 2^60 reject about half / three quarters / 15 in 16 of the candidates: the compaction path
    mods = big + [(1 << 59) + 12345, (1 << 58) + 99, 1 << 56, 1 << 60]
    with avp.VP(spm_rows=len(mods) * n // 128, ksk_rows=1, max_log_n=logn) as vp:
        vp.fill_synthetic(SPM, 0, logn, mods, seed_base=0xA10A, first_index=40)
        got = vp.read(SPM, 0, len(mods) * n).reshape(len(mods), n)
        for i, q in enumerate(mods):
            assert np.array_equal(got[i], splitmix_residues(0xA10A + 40 + i, n, q)), f"limb {i} modulus {q:#x}"
            assert int(got[i].max()) < q
        with pytest.raises(avp.AlohaError):
            vp.fill_synthetic(SPM, 0, logn, [(1 << 60) + 1])
        with pytest.raises(avp.AlohaError):
            vp.fill_synthetic(SPM, 0, logn, [(1 << 56) - 1])         # acceptance rate too low: refused, not looped on
        with pytest.raises(avp.AlohaError):
            vp.fill_synthetic(SPM, 1, logn, mods)                # one row too far


def test_valu_clock_is_a_plausible_mi355x_clock():
    with avp.VP(spm_rows=1 << 10, ksk_rows=1, max_log_n=13) as vp:
        ghz = vp.valu_clock_ghz(30.0)
        assert 1.0 < ghz < 2.6, ghz                              # peak engine clock 2.4 GHz; throttles under multiply load


def test_handle_runs_on_its_own_device_whatever_is_current():
    """entry points used to run on the calling thread's current device.  With two GPUs visible the handle
    lives on device 1 while device 0 is current; with one, the guard must at least leave the current device alone."""
    import torch
    import oracle
    dev = 1 if torch.cuda.device_count() > 1 else 0
    torch.cuda.set_device(0)
    q = avp.nth_prime(13, 0)
    a = np.random.default_rng(3).integers(0, q, (4, 8192), dtype=np.uint64)
    with avp.VP(spm_rows=1 << 10, ksk_rows=1, max_log_n=13, device=dev) as vp:
        assert torch.cuda.current_device() == 0
        vp.write(SPM, 0, a.reshape(-1))
        vp.ntt_rows(SPM, 0, 0, 13, [q] * 4)
        got = vp.read(SPM, 0, 4 * 8192).reshape(4, 8192)
        assert torch.cuda.current_device() == 0
    for i in range(4):
        assert np.array_equal(got[i], oracle.ntt(a[i], q))

