"""
Drop-in mirror of the reference's ``mask_dataloader.py`` (mask_dataloader.py:5-25): every readable image of a
directory as a grayscale int tensor keyed by its base name.  OpenCV / Pillow are used when installed; otherwise
non-interlaced PNG (every colour type and bit depth), binary PGM and ``.npy`` arrays are decoded by the code below.
"""
import os
import struct
import zlib

import numpy as np
import torch


def _unfilter(raw, h, stride, bpp):
    """PNG scanline filters (RFC 2083 §6): None / Up are whole-row numpy operations, Sub is a running sum per byte lane
    (cumsum mod 256); Average and Paeth depend on the byte just reconstructed and run as a tight bytearray loop."""
    out = np.zeros((h, stride), dtype=np.uint8)
    zero = np.zeros(stride, np.uint8)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:]
        prev = out[y - 1] if y else zero
        if ft == 0:
            out[y] = line
        elif ft == 2:
            out[y] = line + prev                                    # uint8 arithmetic wraps mod 256
        elif ft == 1:
            lanes = line.reshape(-1, bpp).astype(np.uint32)
            out[y] = (np.cumsum(lanes, axis=0) & 255).astype(np.uint8).reshape(-1)
        elif ft in (3, 4):
            cur, ln, pv = bytearray(stride), bytes(line), bytes(prev)
            if ft == 3:
                for x in range(stride):
                    a = cur[x - bpp] if x >= bpp else 0
                    cur[x] = (ln[x] + ((a + pv[x]) >> 1)) & 255
            else:
                for x in range(stride):
                    a = cur[x - bpp] if x >= bpp else 0
                    b = pv[x]
                    c = pv[x - bpp] if x >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    cur[x] = (ln[x] + (a if (pa <= pb and pa <= pc) else (b if pb <= pc else c))) & 255
            out[y] = np.frombuffer(bytes(cur), dtype=np.uint8)
        else:
            raise ValueError("corrupt PNG: filter type %d" % ft)
    return out


def _read_png_gray(path):
    """Non-interlaced PNG of any colour type (gray, gray+alpha, RGB, RGBA, palette) and bit depth (1, 2, 4, 8, 16) as the
    8-bit grayscale image cv2.imread(..., IMREAD_GRAYSCALE) returns: 16-bit samples keep their high byte, sub-byte gray
    samples are scaled to 0..255, colours use OpenCV's 0.299 / 0.587 / 0.114 weights, alpha is ignored."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        return None
    pos, idat, ihdr, plte = 8, b"", None, None
    while pos < len(data):
        (length,), ctype = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        if ctype == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif ctype == b"PLTE":
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif ctype == b"IDAT":
            idat += body
        pos += 12 + length
    w, h, depth, ctype, _, _, interlace = ihdr
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype)
    if channels is None or depth not in (1, 2, 4, 8, 16):
        raise ValueError("unsupported PNG colour type %d / bit depth %d" % (ctype, depth))
    if interlace != 0:
        raise ValueError("interlaced (Adam7) PNG masks are not supported by the built-in decoder: re-save the mask "
                         "non-interlaced or install OpenCV / Pillow")
    bits = channels * depth
    stride, bpp = (w * bits + 7) // 8, max(1, bits // 8)
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + stride)
    rows = _unfilter(raw, h, stride, bpp)
    if depth == 16:
        samples = rows.reshape(h, w * channels, 2)[:, :, 0]         # big-endian: the high byte
    elif depth == 8:
        samples = rows
    else:
        samples = np.unpackbits(rows, axis=1)[:, :w * channels * depth].reshape(h, w * channels, depth)
        samples = (samples * (1 << np.arange(depth - 1, -1, -1))).sum(axis=2).astype(np.uint8)
        if ctype == 0:
            samples = (samples.astype(np.uint32) * 255 // ((1 << depth) - 1)).astype(np.uint8)
    img = samples.reshape(h, w, channels)
    if ctype == 3:
        if plte is None:
            raise ValueError("palette PNG without a PLTE chunk")
        img = plte[img[:, :, 0]]
    if img.shape[2] in (1, 2):
        return np.ascontiguousarray(img[:, :, 0])
    rgb = img[:, :, :3].astype(np.float64)
    return np.round(0.299 * rgb[:, :, 0] + 0.587 * rgb[:, :, 1] + 0.114 * rgb[:, :, 2]).astype(np.uint8)


def _read_pgm(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"P5":
        return None
    tok, pos = [], 2
    while len(tok) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos)
            continue
        start = pos
        while not data[pos:pos + 1].isspace():
            pos += 1
        tok.append(int(data[start:pos]))
    w, h, _ = tok
    return np.frombuffer(data[pos + 1:pos + 1 + w * h], dtype=np.uint8).reshape(h, w)


IMAGE_EXTENSIONS = {".png", ".pgm", ".npy", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".webp", ".ppm"}


def _decode(file_path):
    """Grayscale uint8 image or None.  OpenCV first when it is installed (then this IS the reference's loader), Pillow
    next, the built-in PNG / PGM / npy decoders last."""
    ext = os.path.splitext(file_path)[1].lower()
    if ext == ".npy":
        return np.load(file_path)
    try:
        import cv2
        img = cv2.imread(file_path, cv2.IMREAD_GRAYSCALE)
        if img is not None:
            return img
    except ImportError:
        pass
    if ext == ".pgm":
        return _read_pgm(file_path)
    if ext == ".png":
        try:
            return _read_png_gray(file_path)
        except ValueError:
            pass                                    # e.g. interlaced: Pillow may still read it
    try:
        from PIL import Image
        with Image.open(file_path) as im:
            return np.asarray(im.convert("L"))
    except ImportError:
        pass
    except Exception:
        return None
    return _read_png_gray(file_path) if ext == ".png" else None


def load_image_masks(directory_path):
    """mask_dataloader.py:5-25.  One difference, on purpose: a file that LOOKS like an image (by extension) but cannot
    be decoded raises instead of printing a warning -- the reference's warning leaves the camera unmasked, which silently
    changes colours and culling."""
    image_masks = {}
    for filename in sorted(os.listdir(directory_path)):
        file_path = os.path.join(directory_path, filename)
        if os.path.isdir(file_path):
            continue
        ext = os.path.splitext(filename)[1].lower()
        img = _decode(file_path)
        if img is not None:
            image_masks[str(os.path.basename(file_path).split('.')[0])] = torch.tensor(np.ascontiguousarray(img)).to(torch.int)
        elif ext in IMAGE_EXTENSIONS:
            raise ValueError(f"Could not decode mask {file_path}: install OpenCV or Pillow, or convert it to a "
                             f"non-interlaced PNG / binary PGM / .npy")
        else:
            print(f"WARNING: Could not load mask with name {filename}")
    return image_masks
