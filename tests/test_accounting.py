"""CPU checks of the measurement arithmetic (no GPU): the algorithmic-byte formulas behind bench.py's roofline fractions."""
def test_algorithmic_byte_formulas_are_the_surveys():
    """bench.py's roofline fractions of configs 2-5 divide these byte counts by the measured time: pinned here against
    SURVEY.md section 8d evaluated by hand (no GPU)."""
    from aloha_amd import accounting as acc
    n = 1 << 16
    assert acc.ntt_bytes(n) == 1048576                                             # "N=2^16 => 1 048 576 B/NTT"
    assert acc.polymul_bytes(1 << 14, 1) == {"fused_minimum": 384 * 1024, "isa_materialised": 1152 * 1024}   # "384 KiB", "1.125 MiB"
    # the reference's decomposition at L = 24, K = 1: [L + L(L+K) + 2(K+L)] 16N + 2 L (L+K) 8N
    r = acc.keyswitch_bytes(n, 24, 1, 24, "rotate")
    assert r["survey_transforms"] == 24 + 24 * 25 + 2 * 25 == 674 and r["key_limbs"] == 2 * 24 * 25 == 1200
    assert r["survey_total"] == 674 * 16 * n + 1200 * 8 * n
    # hybrid alpha = 4, K = 4 at L = 24 (dnum 6) and L = 32 (dnum 8)
    h = acc.keyswitch_bytes(n, 24, 4, 6, "multiply")
    assert (h["survey_transforms"], h["lean_transforms"], h["key_limbs"], h["io_limbs"]) == (24 + 168 + 56, 24 + 144 + 56, 336, 144)
    assert h["survey_total"] == 248 * 16 * n + 336 * 8 * n and h["lean_total"] == 224 * 16 * n + 336 * 8 * n + 144 * 8 * n
    g = acc.keyswitch_bytes(n, 32, 4, 8, "rotate")
    assert (g["survey_transforms"], g["lean_transforms"], g["key_limbs"], g["io_limbs"]) == (32 + 288 + 72, 32 + 256 + 72, 576, 128)
