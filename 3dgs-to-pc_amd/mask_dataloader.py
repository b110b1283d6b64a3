"""
Drop-in mirror of the reference's ``mask_dataloader.py`` (mask_dataloader.py:5-25): every readable image of a
directory as a grayscale int tensor keyed by its base name.  No OpenCV: 8-bit non-interlaced PNG (gray, gray+alpha,
RGB, RGBA; RGB -> gray with OpenCV's 0.299/0.587/0.114 weights), binary PGM and ``.npy`` arrays are decoded here.
"""
import os
import struct
import zlib

import numpy as np
import torch


def _read_png_gray(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        return None
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        (length,), ctype = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        if ctype == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif ctype == b"IDAT":
            idat += body
        pos += 12 + length
    w, h, depth, ctype, _, _, interlace = ihdr
    channels = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if depth != 8 or interlace != 0 or channels is None:
        return None
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * channels)
    out = np.zeros((h, w * channels), dtype=np.uint8)
    bpp = channels
    for y in range(h):
        ft, line = raw[y, 0], raw[y, 1:].astype(np.int32)
        prev = out[y - 1].astype(np.int32) if y else np.zeros(w * channels, np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros_like(line)
            for x in range(w * channels):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 255
        out[y] = cur
    img = out.reshape(h, w, channels)
    if channels in (1, 2):
        return img[:, :, 0]
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


def load_image_masks(directory_path):
    image_masks = {}
    for filename in os.listdir(directory_path):
        file_path = os.path.join(directory_path, filename)
        try:
            ext = os.path.splitext(filename)[1].lower()
            img = np.load(file_path) if ext == ".npy" else (_read_pgm(file_path) if ext == ".pgm" else _read_png_gray(file_path))
            if img is not None:
                image_masks[str(os.path.basename(file_path).split('.')[0])] = torch.tensor(np.ascontiguousarray(img)).to(torch.int)
            else:
                print(f"WARNING: Could not load mask with name {filename}")
        except Exception as e:
            print(f"ERROR loading mask with name {filename}: {e}")
    return image_masks
