"""Which form stage 1 takes for a batch (csrc/form_select.h, through the host-only hook mj_debug_stage1_form): the rule as a
table of cases — segment count x length x tables x traits -> form, chunk size — without a GPU."""
import numpy as np
import pytest

from pyjpegdecoder_amd import _binding as B

WAVE, LANES, SYNC, SCANS, WG = B.MJ_FORM_WAVE, B.MJ_FORM_LANES, B.MJ_FORM_SYNC, B.MJ_FORM_SCANS, B.MJ_FORM_WG_TABLES
T_BOTH, T_UNORDERED, T_PROG, T_GENERIC, T_GPUSEG, T_ONESEG, T_DCBIG, T_NOSYNC, T_WGFAIL = 1, 2, 4, 8, 16, 32, 64, 128, 256


def segs(n, length):
    return np.full(n, length, dtype=np.int32)


CASES = [
    # name, segment lengths, kwargs, expected form, expected chunk bytes (None = do not care)
    ("config 3: 1024 x 1080p, one MCU row per segment", segs(69632, 9800), {}, LANES, None),
    ("one small file, three segments", segs(3, 700), {}, WAVE, None),
    ("999 short segments: below the lane form's crossover", segs(999, 900), {}, WAVE, None),
    ("1024 short segments: lanes (too short to cut)", segs(1024, 900), {}, LANES, None),
    ("one 1080p file with a restart interval per row: too few lanes, chunks fill the chip", segs(68, 9800), {}, SYNC, 256),
    ("128 such files", segs(68 * 128, 9800), {}, SYNC, 512),
    ("300 such files: the lane form again", segs(68 * 300, 9800), {}, LANES, None),
    ("one 1080p file without restart markers", segs(1, 660000), {}, SYNC, 256),
    ("64 of them", segs(64, 660000), {}, SYNC, 512),
    ("300 of them", segs(300, 660000), {}, SYNC, 1024),
    ("1024 of them: fewer, longer chunks", segs(1024, 660000), {}, SYNC, 2048),
    ("one 24-megapixel file: no more than ~12 000 chunks per segment", segs(1, 10_000_000), {}, SYNC, 1024),
    ("a batch of restart segments with ONE long segment among them", np.concatenate([segs(69632, 9800), segs(1, 70000)]), {}, SYNC, None),
    ("files without markers, NO_SYNC asked for (the serial fallback)", segs(256, 660000), dict(traits=T_NOSYNC), WAVE, None),
    ("... in a batch large enough for lanes", segs(2000, 40000), dict(traits=T_NOSYNC), LANES, None),
    ("a table in both roles: wave form whatever the size", segs(69632, 9800), dict(traits=T_BOTH), WAVE, None),
    ("segments not in blob order", segs(69632, 9800), dict(traits=T_UNORDERED), WAVE, None),
    ("unusual sampling layout", segs(69632, 9800), dict(traits=T_GENERIC), WAVE, None),
    ("progressive", segs(10240, 60000), dict(traits=T_PROG), SCANS, None),
    ("a stream beyond 32-bit offsets", segs(69632, 9800), dict(blob_len=(1 << 32) + 5), WAVE, None),
    ("GPU marker scan, restart segments: lengths unknown, no chunks", segs(68 * 16, 0), dict(traits=T_GPUSEG), LANES, None),
    ("GPU marker scan, one segment per image: the byte range bounds it", segs(64, 670000), dict(traits=T_GPUSEG | T_ONESEG), SYNC, 512),
    ("GPU marker scan, few restart segments", segs(68, 0), dict(traits=T_GPUSEG), WAVE, None),
    ("a DC size above 15: the counting rounds' format has no place for it", segs(256, 660000), dict(traits=T_DCBIG), WAVE, None),
    ("67 tables (optimised files), restart segments, lists fit", segs(68 * 256, 9800), dict(n_huff=67), SYNC | WG, None),
    ("... a full batch", segs(68 * 1024, 9800), dict(n_huff=67), LANES | WG, None),
    ("... lists do not fit (many small files per workgroup)", segs(68 * 1024, 900), dict(n_huff=400, traits=T_WGFAIL), WAVE, None),
    ("... no markers: chunks with per-workgroup lists", segs(256, 660000), dict(n_huff=70), SYNC | WG, None),
    ("forced: wave", segs(69632, 9800), dict(force="wave"), WAVE, None),
    ("forced: lanes on a small batch", segs(12, 800), dict(force="lanes"), LANES, None),
    ("forced: lanes11 on a small batch", segs(12, 800), dict(force="lanes11"), LANES, None),
    ("forced: lanes cannot override a table in both roles", segs(12, 800), dict(force="lanes", traits=T_BOTH), WAVE, None),
    ("forced: sync on the benchmark", segs(69632, 9800), dict(force="sync"), SYNC, 2048),
    ("forced: sync with a forced chunk", segs(64, 660000), dict(force="sync", forced_chunk=4096), SYNC, 4096),
    ("forced: lanes keeps files without markers out of the chunks", segs(2000, 660000), dict(force="lanes"), LANES, None),
    ("empty batch", segs(0, 0), {}, WAVE, None),
]


@pytest.mark.parametrize("name,seg_len,kw,form,chunk", CASES, ids=[c[0] for c in CASES])
def test_stage1_form_rule(name, seg_len, kw, form, chunk):
    got_form, got_chunk, n_chunks, _ = B.stage1_form_rule(seg_len, **kw)
    assert got_form == form, (name, got_form)
    if chunk is not None:
        assert got_chunk == chunk, (name, got_chunk)
        assert n_chunks == sum(max(1, -(-int(l) // got_chunk)) for l in seg_len)


def test_chunk_size_keeps_the_batch_under_its_chunk_budget():
    """The shortest of 256 / 512 / 1024 / 2048 bytes that keeps the batch under ~50 000 chunks (256) or ~330 000 (the others) and a
    segment under ~12 000."""
    budget = lambda chunk: 50000 if chunk == 256 else 330000
    for n, length in ((1, 200_000), (16, 660_000), (20, 660_000), (64, 660_000), (256, 660_000), (300, 660_000), (1024, 660_000), (1, 5_000_000),
                      (1, 13_000_000)):
        _, chunk, n_chunks, _ = B.stage1_form_rule(segs(n, length))
        assert chunk in (256, 512, 1024, 2048)
        if chunk > 256:
            assert n * length // (chunk // 2) > budget(chunk // 2) or length // (chunk // 2) > 12000
        if chunk < 2048:
            assert n * length // chunk <= budget(chunk) and length // chunk <= 12000


def test_segments_are_dealt_out_by_length_only_when_they_differ():
    rng = np.random.default_rng(0)
    even = rng.integers(9000, 10500, 69632).astype(np.int32)
    assert not B.stage1_form_rule(even)[3]
    mixed = even.copy()
    mixed[::50] = 30000
    assert B.stage1_form_rule(mixed)[3]
    assert not B.stage1_form_rule(segs(10, 5000))[3]


def test_fused_launch_shapes():
    """How a fused launch is cut (fused.hip: fused_shape): whole images per workgroup, 8 producer wavefronts of at most 64 lanes,
    as many consumers as the LDS left beside them holds (8 320 B each for 4:2:0) — every workgroup resident at once where the
    batch fits the chip's lanes, and in PASSES (a workgroup's images a few at a time) where it does not."""
    c3 = B.fused_shape_rule(1024, 68)
    assert c3 == dict(ok=True, images_per_wg=4, producers=8, lanes=34, consumers=8, producer_lds=c3["producer_lds"], passes=1, workgroups=256)
    assert c3["producer_lds"] <= 92 * 1024
    c4 = B.fused_shape_rule(1250, 68)                   # config 4's share: 5 images per workgroup, 340 lanes
    assert (c4["ok"], c4["images_per_wg"], c4["producers"], c4["lanes"], c4["passes"]) == (True, 5, 8, 43, 1) and 4 <= c4["consumers"] < 8
    half = B.fused_shape_rule(512, 68)
    assert (half["images_per_wg"], half["producers"], half["lanes"], half["consumers"], half["workgroups"]) == (2, 4, 34, 8, 256)
    one = B.fused_shape_rule(200, 68)
    assert (one["images_per_wg"], one["producers"], one["lanes"], one["workgroups"]) == (1, 2, 34, 200)
    # more segments than 8 x 64 lanes per workgroup: round 5 refused these, now the workgroup takes its images in passes
    two = B.fused_shape_rule(2048, 68)                  # 8 images per workgroup = 544 segments: 2 passes of 4
    assert (two["ok"], two["images_per_wg"], two["passes"], two["workgroups"], two["lanes"]) == (True, 4, 2, 256, 34)
    uhd = B.fused_shape_rule(1024, 135, hmax=2, vmax=1)  # 1080p 4:2:2 (or 2160p 4:2:0) in rows: 4 images = 540 segments: 2 passes of 2
    assert (uhd["ok"], uhd["images_per_wg"], uhd["passes"], uhd["workgroups"], uhd["producers"] * uhd["lanes"] >= 270) == (True, 2, 2, 256, True)
    halfrow = B.fused_shape_rule(1024, 270, hmax=1, vmax=1)   # 1080p 4:4:4, restart interval half an MCU row: one image per pass, four passes
    assert (halfrow["ok"], halfrow["images_per_wg"], halfrow["passes"], halfrow["workgroups"]) == (True, 1, 4, 256)
    assert not B.fused_shape_rule(64, 513)["ok"]        # one image's segments must fit the producers' lanes
    assert not B.fused_shape_rule(200000, 68)["ok"]     # (more than 64 passes: the two launches)
    assert B.fused_shape_rule(1024, 68, want_consumers=3)["consumers"] == 3
    assert not B.fused_shape_rule(1024, 68, want_consumers=0)["ok"]
    # three 13-bit-sized tables would not leave room: the budget is what decides
    assert not B.fused_shape_rule(1024, 68, n_ac=3, ac_slot_bytes=37376)["ok"]
    # row-major: the strip worker's geometry is the transposed image's (4:2:2 becomes 4:4:0 and back)
    a, b = B.fused_shape_rule(1000, 45, hmax=2, vmax=1, transposed=True), B.fused_shape_rule(1000, 45, hmax=1, vmax=2)
    assert a == b and a["ok"]
    for n, spi in ((1024, 68), (1250, 68), (777, 30), (1021, 60), (64, 17), (3000, 68), (1500, 135), (5, 400)):
        s = B.fused_shape_rule(n, spi)
        if s["ok"]:
            assert s["images_per_wg"] * s["passes"] * s["workgroups"] >= n and s["workgroups"] <= 256
            assert s["producers"] * s["lanes"] >= s["images_per_wg"] * spi and s["images_per_wg"] * spi <= 512
            assert s["producers"] + s["consumers"] <= 16 and s["lanes"] <= 64
            # no workgroup without work, no pass more than needed
            assert (s["workgroups"] - 1) * s["passes"] * s["images_per_wg"] < n


def test_progressive_split_tiers():
    """Which refining AC scans of a progressive batch are walked as scout + parts (form_select.h: choose_prog_split): all of them
    while the chip has wave slots for the extra walks, then each image's largest with two parts, then none — libjpeg's 10-scan
    script on 1080p files (one segment per scan), the family the thresholds were measured on (384 .. 1536 files)."""
    def batch(n, sizes=(75000, 72000, 110000, 290000)):       # refining AC scans per image (bytes): Cb, Cr, luma's first refinement, luma's last
        scans = []
        for i in range(n):
            scans += [(i, 1, -1)] * 5                            # DC, three first AC scans, DC refinement ... : never candidates
            scans += [(i, 1, b) for b in sizes]
            scans += [(i, 1, -1)]
        return scans
    cand = [5, 6, 7, 8]
    # 8 192 wave slots: four scouts + 16 parts per image fit four fifths of them up to 327 images, three walks per image two fifths up to 1 092
    for n, want in ((16, "all"), (327, "all"), (328, "largest"), (768, "largest"), (1024, "largest"), (1092, "largest"), (1093, "none"), (2048, "none")):
        split, parts = B.prog_split_rule(n, batch(n))
        per_image = [[split[i * 10 + c] for c in cand] for i in (0, n - 1)]
        if want == "all":
            assert per_image == [[True] * 4] * 2 and parts == 4, (n, per_image)
            assert sum(split) == 4 * n
        elif want == "largest":
            assert per_image == [[False, False, False, True]] * 2 and parts == 2, (n, per_image)     # the last luma refinement
        else:
            assert not any(split) and parts == 4, n
    # a band's walk must be long enough: 68 bands x 1 KiB
    split, _ = B.prog_split_rule(4, batch(4, sizes=(69631, 69632, 1000, 200000)))
    assert [split[c] for c in cand] == [False, True, False, True]
    # the switches: never / every candidate / the largest of each image; a caller's parts are kept
    assert not any(B.prog_split_rule(4, batch(4), mode=0)[0])
    assert sum(B.prog_split_rule(2048, batch(2048, sizes=(10, 20, 30, 40)), mode=2)[0]) == 4 * 2048
    split, parts = B.prog_split_rule(2048, batch(2048), mode=3, parts=3)
    assert sum(split) == 2048 and parts == 3 and split[8]
    assert B.prog_split_rule(700, batch(700), parts=8)[1] == 8
    # a smaller chip runs out of slots sooner
    small = B.prog_split_rule(200, batch(200), wave_slots=64 * 32)[0]
    assert small[8] and not small[5]


def test_which_batches_take_the_fused_launch():
    """form_select.h: fused_applies — the rule in front of the fused launch (mj_plan_create asks it, then fused_shape for the LDS):
    uniform colour batches of the common samplings in the resolved-table lane form, interleaved pixels, no seam outputs; the
    restart interval decides by pixel layout (x-major: a divisor of the MCU row, or two rows — measured, EXPERIMENTS.md;
    row-major: any), at most 512 segments per image."""
    X, R, PX = B.MJ_LAYOUT_XMAJOR, B.MJ_LAYOUT_ROWMAJOR, B.MJ_LAYOUT_PLANAR_XMAJOR
    rule = B.fused_applies_rule
    # 1080p 4:2:0: 120 x 68 MCUs
    assert rule(X, 2, 2, 120, 68, 120) == 1 and rule(R, 2, 2, 120, 68, 120) == 1
    for ri, x_major in ((60, 1), (40, 1), (24, 1), (240, 1), (360, 0), (100, 0), (50, 0), (480, 0)):
        assert rule(X, 2, 2, 120, 68, ri) == x_major, ri
        assert rule(R, 2, 2, 120, 68, ri) == 1, ri
    assert rule(X, 2, 2, 120, 68, 0) == 0                       # no restart markers: the synchronisation form's
    assert rule(X, 2, 2, 120, 68, 15) == 0 and rule(R, 2, 2, 120, 68, 15) == 0     # 544 segments per image: more than a workgroup's lanes
    assert rule(X, 2, 2, 120, 68, 16) == 0 and rule(R, 2, 2, 120, 68, 16) == 1     # 510 fit; 16 does not divide 120
    # samplings: 4:4:4 / 4:2:2 / 4:4:0 both ways, 4:1:1 in x-major output only, nothing outside those
    for h, v in ((1, 1), (2, 1), (1, 2)):
        assert rule(X, h, v, 120, 68, 120) == 1 and rule(R, h, v, 120, 68, 120) == 1
    assert rule(X, 4, 1, 60, 135, 60) == 1 and rule(X, 4, 1, 60, 135, 120) == 1 and rule(R, 4, 1, 60, 135, 60) == 0
    assert rule(X, 1, 4, 240, 34, 240) == 0 and rule(X, 4, 2, 60, 68, 60) == 0
    assert rule(X, 1, 1, 240, 135, 240, ncomp=1) == 0           # greyscale
    # files of mixed content (segments dealt out by length): the form with the hand-off across workgroups
    assert rule(X, 2, 2, 120, 68, 120, traits=2) == 2 and rule(R, 2, 2, 120, 68, 60, traits=2) == 2
    for traits in (1, 4, 8, 16, 32, 64):                        # not the lane form / another order / several geometries / generic / progressive / intervals differ
        assert rule(X, 2, 2, 120, 68, 120, traits=traits) == 0, traits
    assert rule(PX, 2, 2, 120, 68, 120) == 0                    # planar pixels
    for flags in (B.MJ_FLAG_EXACT_ONLY, B.MJ_FLAG_KEEP_PLANES, B.MJ_FLAG_KEEP_IDCT):
        assert rule(X, 2, 2, 120, 68, 120, flags=flags) == 0
    assert rule(X, 2, 2, 120, 68, 120, flags=B.MJ_FLAG_GPU_SEGMENT) == 1

