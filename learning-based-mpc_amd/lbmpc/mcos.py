"""Decode MPT3 `Polyhedron` objects (MATLAB MCOS opaque objects) out of the reference's .mat
fixtures with scipy only (nothing from the file is executed).  Recipe: SURVEY.md Appendix A.

Part of the .mat I/O of the package (lbmpc/matio.py: SURVEY.md 8f-4): a user of the reference points the library at
the parameters_*.mat / Results_*.mat files they already have.  tests/golden/make_fixtures.py uses it in the build
container to turn the reference's stored data into the committed .npz fixtures.
"""
import struct
from io import BytesIO

import numpy as np
import scipy.io as sio
from scipy.io.matlab._mio5 import MatFile5Reader


class McosFile:
    def __init__(self, path):
        self.mat = sio.loadmat(path, struct_as_record=True, squeeze_me=False)
        self.arr = None
        if "__function_workspace__" in self.mat:
            self._parse_workspace(self.mat["__function_workspace__"].tobytes())

    def _parse_workspace(self, raw):
        hdr = bytearray(128)
        tag = b"MATLAB 5.0 MAT-file"
        hdr[: len(tag)] = tag
        hdr[124:126] = raw[0:2]
        hdr[126:128] = raw[2:4]
        rdr = MatFile5Reader(BytesIO(bytes(hdr) + raw[8:]), struct_as_record=True, squeeze_me=False)
        rdr.initialize_read()
        rdr.mat_stream.seek(128)
        vh, _ = rdr.read_var_header()
        ws = rdr.read_var_array(vh, process=False)
        self.arr = ws["MCOS"][0, 0][0]["arr"]
        meta = self.arr[0, 0].tobytes()
        _, nstr = struct.unpack_from("<II", meta, 0)
        offs = struct.unpack_from("<8I", meta, 8)
        names, p = [], 40
        for _ in range(nstr):
            e = meta.index(b"\0", p)
            names.append(meta[p:e].decode("ascii"))
            p = e + 1
        self.names = names
        cls = np.frombuffer(meta[offs[0]:offs[1]], dtype="<u4").reshape(-1, 4)
        self.class_names = [names[c[1] - 1] if c[1] else "" for c in cls]
        self.objs = np.frombuffer(meta[offs[2]:offs[3]], dtype="<u4").reshape(-1, 6)
        self.seg2 = self._props(meta[offs[1]:offs[2]])
        self.seg4 = self._props(meta[offs[3]:offs[4]])

    @staticmethod
    def _props(seg):
        out, p = [], 8
        while p + 4 <= len(seg):
            (n,) = struct.unpack_from("<I", seg, p)
            p += 4
            trip = [struct.unpack_from("<3I", seg, p + 12 * i) for i in range(n)]
            p += 12 * n
            p += (-p) % 8
            out.append(trip)
        return out

    def object_props(self, opaque):
        """opaque: the numpy record scipy returns for an MCOS object -> list of dict(prop->value)."""
        ref = np.asarray(opaque["arr"]).ravel()
        if ref.dtype == object:
            ref = np.asarray(ref[0]).ravel()
        ref = ref.astype(np.uint32)
        assert ref[0] == 0xDD000000, hex(int(ref[0]))
        nd = int(ref[1])
        dims = ref[2:2 + nd]
        n = int(np.prod(dims))
        ids = ref[2 + nd: 2 + nd + n]
        res = []
        for oid in ids:
            row = self.objs[int(oid)]
            plist = self.seg4[int(row[4]) - 1] if row[4] else self.seg2[int(row[3]) - 1]
            d = {}
            for name_idx, flag, val in plist:
                nm = self.names[name_idx - 1]
                if flag == 1:
                    d[nm] = self.arr[val + 2, 0]
                elif flag == 0:
                    d[nm] = self.names[val - 1]
                else:
                    d[nm] = val
            res.append(d)
        return res

    def polyhedron(self, opaque):
        """-> dict(H=[A b] or None, V or None)"""
        d = self.object_props(opaque)[0]
        H = np.asarray(d.get("H_int", np.zeros((0, 0))), dtype=float)
        V = np.asarray(d.get("V_int", np.zeros((0, 0))), dtype=float)
        return {"H": H if H.size else None, "V": V if V.size else None}
