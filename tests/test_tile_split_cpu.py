"""CPU tier: host-side arithmetic the pixel kernel relies on (tests/hostsim builds the same headers for the CPU)."""


def test_tile_index_split_by_magic_multiply_is_exact():
    """The pixel kernel splits a tile index into tile row and column with a multiply by floor(2^32 / tiles_x) and one correction
    (RowMap::tiles_x_magic, sdfr_frame.h) instead of a division: exact for every frame width and tile index a launch can have."""
    import ctypes
    import random

    import hostsim

    fn = hostsim.lib().hostsim_tile_split
    fn.restype = ctypes.c_ulonglong
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint]
    rng = random.Random(11)
    widths = [1, 2, 7, 8, 9, 63, 64, 65, 200, 256, 1920, 3840, 7680, 16384, 65535] + [rng.randrange(1, 40000) for _ in range(60)]
    for width in widths:
        for tw in (3, 4, 5, 6):
            tiles_x = (width + (1 << tw) - 1) >> tw
            tiles = [0, 1, tiles_x - 1, tiles_x, tiles_x + 1, 2 * tiles_x - 1, 2 * tiles_x, 0xffffffff, 0xfffffffe, 0x7fffffff, 0x80000000]
            tiles += [rng.randrange(0, 1 << 32) for _ in range(40)] + [rng.randrange(0, 1 << 24) for _ in range(40)]
            for t in tiles:
                if t < 0:
                    continue
                v = fn(width, tw, t)
                assert (v >> 32, v & 0xffffffff) == (t // tiles_x, t % tiles_x), (width, tw, t)
