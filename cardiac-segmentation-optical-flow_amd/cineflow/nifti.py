"""Minimal NIfTI-1 (.nii / .nii.gz) reader and writer.

The reference does its file I/O through SimpleITK (`sitk.ReadImage` in nnunet/preprocessing/cropping.py:75-89,
`sitk.WriteImage` in nnunet/inference/segmentation_export.py:190-219); SimpleITK and nibabel are absent from this
image and nothing may be installed, so the predict-from-folder shell uses this small codec instead.  Arrays follow the
SimpleITK convention the reference's properties dict uses: `array[z, y, x]` with `itk_spacing = (sx, sy, sz)`,
`itk_origin = (ox, oy, oz)` and a row-major 3x3 `itk_direction`, all in ITK's LPS frame (NIfTI stores RAS: the x and y
axes flip sign on the way in and out).
"""
import gzip
import struct
import zlib

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


# zlib level of the .nii.gz files written here.  Python's gzip default (9) spent 0.42 s per 0.5 MB label map of the API bench -- the whole
# export was compression (profiles/r03_api_split.md); ITK's NIfTI writer, which the reference goes through, compresses at a low level too.  The
# level changes the file size only: the voxels a reader gets back are identical.
GZIP_LEVEL = 2


def _gunzip(raw):
    """All members of a gzip stream, one decompressobj call per member.  Python 3.10's gzip.decompress -- and zlib.decompress with its default
    16 KB output buffer -- inflate in ~110 small steps and take the GIL back after each: eight reader threads ran one after the other
    (1.13 s for 80 volumes of 1.8 MB; this form 0.23 s, a single thread 0.13 s per 10)."""
    out = []
    while raw:
        d = zlib.decompressobj(wbits=31)
        out.append(d.decompress(raw))
        if not d.eof:
            raise ValueError("truncated gzip stream")
        raw = d.unused_data.lstrip(b"\0")          # (zero padding between / after members is legal)
    return out[0] if len(out) == 1 else b"".join(out)


def _gzip(data, level):
    """gzip container in one zlib call (gzip.compress goes through GzipFile.write and scales 2.2x on 8 threads, this 6.5x)"""
    c = zlib.compressobj(level, zlib.DEFLATED, 31)
    return c.compress(data) + c.flush()


def _read_all(path):
    with open(path, "rb") as f:
        raw = f.read()
    return _gunzip(raw) if str(path).endswith(".gz") else raw


def _write_all(path, data):
    with open(path, "wb") as f:
        f.write(_gzip(data, GZIP_LEVEL) if str(path).endswith(".gz") else data)


def write_nifti(path, array, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=(1, 0, 0, 0, 1, 0, 0, 0, 1)):
    """array [Z,Y,X] (or [Y,X]); spacing/origin/direction in ITK (x,y,z / LPS) convention."""
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[None]
    assert a.ndim == 3, "only 3-D volumes"
    if a.dtype not in _CODES:
        a = a.astype(np.float32)
    nz, ny, nx = a.shape
    D = np.asarray(direction, dtype=np.float64).reshape(3, 3)
    lps2ras = np.diag([-1.0, -1.0, 1.0])
    A = lps2ras @ D @ np.diag(np.asarray(spacing, dtype=np.float64))
    t = lps2ras @ np.asarray(origin, dtype=np.float64)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into("<h", hdr, 70, _CODES[a.dtype])
    struct.pack_into("<h", hdr, 72, a.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, 1.0, float(spacing[0]), float(spacing[1]), float(spacing[2]), 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<f", hdr, 108, 352.0)          # vox_offset
    struct.pack_into("<f", hdr, 112, 1.0)            # scl_slope
    struct.pack_into("<B", hdr, 123, 2)              # xyzt_units: mm
    struct.pack_into("<h", hdr, 252, 0)              # qform_code
    struct.pack_into("<h", hdr, 254, 1)              # sform_code: scanner
    for r in range(3):
        struct.pack_into("<4f", hdr, 280 + 16 * r, float(A[r, 0]), float(A[r, 1]), float(A[r, 2]), float(t[r]))
    hdr[344:348] = b"n+1\0"
    _write_all(path, bytes(hdr) + b"\0\0\0\0" + np.ascontiguousarray(a).tobytes())


def read_nifti(path):
    """-> (array [Z,Y,X], {'itk_spacing','itk_origin','itk_direction'})."""
    raw = _read_all(path)
    if struct.unpack_from("<i", raw, 0)[0] != 348:
        raise ValueError("%s: not a little-endian NIfTI-1 file" % path)
    dim = struct.unpack_from("<8h", raw, 40)
    code = struct.unpack_from("<h", raw, 70)[0]
    pixdim = struct.unpack_from("<8f", raw, 76)
    vox_offset = int(struct.unpack_from("<f", raw, 108)[0])
    slope, inter = struct.unpack_from("<2f", raw, 112)
    if code not in _DTYPES:
        raise ValueError("%s: unsupported NIfTI datatype %d" % (path, code))
    if dim[0] > 3 and any(d > 1 for d in dim[4:1 + dim[0]]):
        raise ValueError("%s: %d-D NIfTI (dim %s); only 3-D volumes are supported (one file per frame, as the reference's dataset "
                         "conversion writes them)" % (path, dim[0], dim[1:1 + dim[0]]))
    nx, ny, nz = dim[1], max(dim[2], 1), max(dim[3], 1) if dim[0] >= 3 else 1
    n = nx * ny * nz
    a = np.frombuffer(raw, dtype=_DTYPES[code], count=n, offset=max(vox_offset, 352)).reshape(nz, ny, nx)
    if slope not in (0.0, 1.0) or inter != 0.0:
        a = a.astype(np.float32) * (slope if slope != 0.0 else 1.0) + inter
    sform_code = struct.unpack_from("<h", raw, 254)[0]
    spacing = np.array([pixdim[1] or 1.0, pixdim[2] or 1.0, pixdim[3] or 1.0], dtype=np.float64)
    lps2ras = np.diag([-1.0, -1.0, 1.0])
    if sform_code > 0:
        A = np.array([struct.unpack_from("<4f", raw, 280 + 16 * r) for r in range(3)], dtype=np.float64)
        M, t = A[:, :3], A[:, 3]
        sp = np.linalg.norm(M, axis=0)
        sp[sp == 0] = 1.0
        D = lps2ras @ (M / sp)
        origin = lps2ras @ t
        spacing = sp
    elif struct.unpack_from("<h", raw, 252)[0] > 0:
        # qform only (scanner- / ITK-written files): rotation from the quaternion (b, c, d), qfac = pixdim[0] flips the third axis --
        # the fallback ITK itself takes when sform_code is 0
        b, c, d = struct.unpack_from("<3f", raw, 256)
        off = np.array(struct.unpack_from("<3f", raw, 268), dtype=np.float64)
        a2 = 1.0 - (b * b + c * c + d * d)
        qa = np.sqrt(a2) if a2 > 1e-7 else 0.0
        if a2 <= 1e-7:
            nrm = 1.0 / np.sqrt(b * b + c * c + d * d)
            b, c, d = b * nrm, c * nrm, d * nrm
        R = np.array([[qa * qa + b * b - c * c - d * d, 2 * (b * c - qa * d), 2 * (b * d + qa * c)],
                      [2 * (b * c + qa * d), qa * qa + c * c - b * b - d * d, 2 * (c * d - qa * b)],
                      [2 * (b * d - qa * c), 2 * (c * d + qa * b), qa * qa + d * d - b * b - c * c]], dtype=np.float64)
        if pixdim[0] < 0:
            R[:, 2] = -R[:, 2]
        D = lps2ras @ R
        origin = lps2ras @ off
    else:
        D = np.eye(3)
        origin = np.zeros(3)
    props = {"itk_spacing": tuple(float(v) for v in spacing), "itk_origin": tuple(float(v) for v in origin),
             "itk_direction": tuple(float(v) for v in D.reshape(-1))}
    return np.array(a), props
