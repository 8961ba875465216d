"""The LDS-DMA form of the large-brick bf16 forward kernel (csrc/conv3d_bf16.hip, conv3d_fwd_bf16_v2_kernel<.., DMA = true>)
stores 32-byte rows unpadded and swaps the two 16-byte halves of some rows instead.  This test enumerates, for every
ds_read_b128 the kernel issues, the 16-lane groups the LDS serves in one cycle (MI355X_MICROARCH.md, LDS table) and checks
that the sixteen lanes of a group fall into sixteen different 16-byte slots of the 256-byte bank row: conflict-free."""
import itertools

HH = HW = 10
GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def slots(addr_of_lane):
    for g in GROUPS:
        yield sorted((addr_of_lane(l) // 16) % 16 for l in g)


def test_halo_fragment_reads_are_conflict_free():
    """A operand: lane = (plane pd = l & 3, column pw = l >> 2) of brick row `wave`, k half = lane >> 5; tap (kd, kh, kw),
    M-tile m (planes 4 m ..): halo row ((pd + kd + 4 m) * HH + wave + kh) * HW + pw + kw, halves swapped where bit 1 of
    the plane index is set."""
    for wave, kd, kh, kw, m in itertools.product(range(8), range(3), range(3), range(3), range(2)):
        def addr(lane):
            l31, hsel = lane & 31, lane >> 5
            hd = (l31 & 3) + kd + 4 * m
            row = (hd * HH + wave + kh) * HW + (l31 >> 2) + kw
            return row * 32 + ((hsel ^ ((hd >> 1) & 1)) * 16)
        for s in slots(addr):
            assert s == list(range(16)), (wave, kd, kh, kw, m, s)


def test_weight_fragment_reads_are_conflict_free():
    """B operand: lane = output channel l & 31 of N-tile j, k half = lane >> 5; row tap * NB + 32 j + (l & 31), halves
    swapped where bit 3 of the channel is set."""
    for nb, tp, j in itertools.product((32, 64), range(3), range(2)):
        if j * 32 >= nb:
            continue

        def addr(lane):
            l31, hsel = lane & 31, lane >> 5
            return (tp * nb + j * 32 + l31) * 32 + ((hsel ^ ((l31 >> 3) & 1)) * 16)
        for s in slots(addr):
            assert s == list(range(16)), (nb, tp, j, s)


def test_unswizzled_rows_would_conflict():
    """The same reads without the swap put two lanes of a group into one slot (why the register-staged form pads its rows)."""
    def addr(lane):
        l31, hsel = lane & 31, lane >> 5
        return (((l31 & 3) * HH) * HW + (l31 >> 2)) * 32 + hsel * 16
    assert any(len(set(s)) < 16 for s in slots(addr))
