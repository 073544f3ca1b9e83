"""Phase-stamp buffer layout of the GEMV kernels (teal_set_phase_buffer): 32 uint64 per workgroup —
[0] kernel entry, [1] kernel arguments in registers, [2] activation ready, [3] row list ready, [4] first weight batch
consumed (wave 0), [5] wave 0 done streaming, [6] past the reduce barrier, [7] done, [12] hw id << 32 | xcc,
[13] waves << 32 | workgroups, [16 + w] end of stream of wave w.  100 MHz wall clock."""
ROW = 32


def legacy_view(phase, wgs):
    """[wgs, 8] int64 in the column meaning the round-1 scripts use: 0 start, 1 activation ready, 2/3 list ready,
    4 rows streamed (wave 0), 5 done, 6 activation ready, 7 hw id | xcc."""
    p = phase[: wgs * ROW].view(wgs, ROW)
    return p[:, [0, 2, 3, 3, 5, 7, 2, 12]]
