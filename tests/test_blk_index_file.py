"""The library's reader of the reference's block index file (spdp_blk_index_read, spaln_amd/csrc/spdp_blk_index_io.cpp) against
what the reference itself held after opening the same file: tests/golden/blk_k*.bkn are the files `spaln -W` wrote,
blk_k*.spdg carry the arrays and parameters of the reference's SrchBlk object (recorded by oracle/ref_build/blk_tap.cc).
Host only -- no GPU needed."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from spaln_amd import blocks, engine
from tests import spdg

HERE = os.path.dirname(os.path.abspath(__file__))
P = dict(nalpha=0, tabsize=3, nshift=5, blklen=6, nbitpat=8, convts=10, n_chr=12, maxblk=14, kk=15, drna=16, maxmmc=17, nseg=19,
         minsigpr=22, ncand=23, nascr=24, maxblock=25, extblock=26, extblockl=27, shortquery=28, hh_size=29, hh_step=30,
         hb_size=31, hb_step=32, ha_size=33, ha_step=34, gdb=38)


@pytest.mark.parametrize("name", ["blk_k1", "blk_k3", "blk_p1"])         # blk_p1: <db>.bkp of `spaln -W -KP` (amino-acid words of the translated genome)
def test_reader_equals_what_the_reference_held(name):
    lib = C.CDLL(engine.LIB_PATH)
    fx = spdg.load(os.path.join(HERE, "golden", name + ".spdg"))
    prm = np.asarray(fx["blk_prm"])
    # ExtBlock comes from the species' intron length distribution (outside the index): handed over as the caller would
    got = blocks.read_index_file(lib, os.path.join(HERE, "golden", name + (".bkp" if name == "blk_p1" else ".bkn")), ext_block=int(prm[26]), max_out=int(prm[39]))
    for k, pos in P.items():
        assert got[k] == int(prm[pos]), (k, got[k], int(prm[pos]))
    f = lambda i: struct.unpack("<f", struct.pack("<i", int(prm[i])))[0]
    assert got["rbscoef"] == f(36) and got["rbscons"] == f(37)
    assert [got["bclw"], got["bcup"], got["bcce"]] == np.frombuffer(np.asarray(fx["blk_pb2c"], np.uint8).tobytes(), np.float64).tolist()
    for key, dt in (("blk_nblk", np.uint16), ("blk_wscr", np.int16), ("blk_blkp", np.int32), ("blk_blkb", np.uint32),
                    ("blk_rscrtab", np.int32), ("blk_chr", np.int32), ("blk_bitpat", np.int32)):
        assert np.array_equal(got[key], np.asarray(fx[key]).view(dt) if np.asarray(fx[key]).dtype.itemsize == np.dtype(dt).itemsize
                              else np.asarray(fx[key]).astype(dt)), key
    assert np.array_equal(got["blk_convtab"][2:], np.asarray(fx["blk_convtab"], np.uint8)[2:])     # ([0], [1]: never written by the reference)


def test_reader_refuses_what_it_does_not_read(tmp_path):
    lib = C.CDLL(engine.LIB_PATH)
    raw = bytearray(open(os.path.join(HERE, "golden", "blk_k1.bkn"), "rb").read())
    bad = tmp_path / "old.bkn"
    raw[36 + 46:36 + 48] = struct.pack("<H", 25)             # ContBlk::VerNo
    bad.write_bytes(raw)
    with pytest.raises(RuntimeError, match="version 26"):
        blocks.read_index_file(lib, str(bad))
    short = tmp_path / "short.bkn"
    short.write_bytes(open(os.path.join(HERE, "golden", "blk_k1.bkn"), "rb").read()[:5000])
    with pytest.raises(RuntimeError):
        blocks.read_index_file(lib, str(short))
