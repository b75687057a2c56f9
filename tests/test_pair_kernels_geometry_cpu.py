"""CPU: the layout claims of the round-4 pair kernels restated in numpy — no GPU, no library calls.

* kernels/respair_cl_bf16.hip (second form): the x / h tile is stored WITHOUT row padding, 16-byte piece p of row r at piece
  p ^ swizzle(r).  Claim: every ds_read_b128 the GEMM issues (lanes = 32 consecutive rows x the two 8-channel halves of one 16-channel
  group, any tap shift) is bank-conflict-free, i.e. each of the instruction's four 16-lane groups touches 16 distinct 16-byte units of
  the 64-bank (256-byte) LDS word line (MI355X_MICROARCH.md, LDS table: ds_read_b128 = 4 x 16 lanes, bank = (addr / 4) mod 64).
* kernels/respair_cl_bf16.hip / respair_x6.hip: tiles of HT computed columns yield HT - (k - 1) outputs; the tiling covers every output
  row exactly once and the staged window (HT + (k - 1) dil rows, starting (k - 1) / 2 * (dil + 1) before the tile) holds everything
  conv1 reads for the rows of h that conv2 needs."""
import numpy as np

# ds_read_b128 lane groups (guide): {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def swizzle(C, r):
    return ((r >> 1) & 7) if C == 64 else (r & 15)


def byte_addr(C, r, p):
    return r * 2 * C + ((p ^ swizzle(C, r)) << 4)


def test_swizzled_tile_reads_are_bank_conflict_free():
    for C in (64, 128, 256):
        G = C // 16
        for row0 in (0, 1, 5, 17, 128, 255):                   # wave's first row + tap shift: any value
            for s in range(G):                                  # 16-channel group
                lanes = np.arange(64)
                l31, lh = lanes & 31, lanes >> 5
                addr = np.array([byte_addr(C, row0 + int(a), 2 * s + int(b)) for a, b in zip(l31, lh)])
                assert len(set(addr.tolist())) == 64            # every lane its own piece
                for grp in GROUPS:
                    units = (addr[grp] // 16) % 16              # 16-byte unit inside the 256-byte word line
                    assert len(set(units.tolist())) == 16, (C, row0, s, sorted(units.tolist()))


def test_swizzle_is_a_permutation_of_each_row():
    for C in (64, 128, 256):
        ppr = C // 8
        for r in range(64):
            assert sorted((p ^ swizzle(C, r)) for p in range(ppr)) == list(range(ppr))


def test_padded_pitch_of_the_first_form_is_conflict_free_too():
    # pitch C + 8 elements = odd multiple of 16 bytes: 16 consecutive rows hit 16 distinct units
    for C in (32, 64, 128, 256):
        pitch = (C + 8) * 2
        assert (pitch // 16) % 2 == 1
        for grp in GROUPS:
            rows = np.array(grp) & 31
            units = ((rows * pitch) // 16) % 16
            assert len(set(units.tolist())) == 16


def _check_tiling(HT, XR, k, dil, L):
    BT = HT - (k - 1)
    p2 = (k - 1) // 2
    p1 = p2 * dil
    assert HT + (k - 1) * dil <= XR
    covered = np.zeros(L, dtype=int)
    for t0 in range(0, L, BT):
        rows = min(BT, L - t0)
        covered[t0:t0 + rows] += 1
        # outputs t0 .. t0 + rows - 1 read h at times t - p2 .. t + p2: h columns (t - p2 + j) - (t0 - p2) in [0, HT)
        hmin, hmax = 0, (rows - 1) + (k - 1)
        assert hmax < HT
        # h column hc (time t0 - p2 + hc) reads x at times .. - p1 + j * dil: staged columns hc + j * dil in [0, HT + (k-1) dil)
        assert hmax + (k - 1) * dil < HT + (k - 1) * dil <= XR
        assert (t0 - p2) - p1 == t0 - p2 - p1                    # first staged time step
    assert (covered == 1).all()


def test_pair_tilings_cover_every_output_once():
    for k in (3, 7, 11):
        for dil in (1, 3, 5):
            for L in (1, 117, 246, 247, 3072, 24576 + 5):
                _check_tiling(256, 320, k, dil, L)               # respair_x6 C <= 64, respair_cl_bf16 form 1 at C >= 128 (XR = rows allocated)
                _check_tiling(128, 192, k, dil, L)               # respair_x6 C = 128, respair_cl_bf16 form 0 at C >= 128
                _check_tiling(512, 512 + 50, k, dil, L)          # respair_cl_bf16: C = 64 form 1, C = 32 form 0
