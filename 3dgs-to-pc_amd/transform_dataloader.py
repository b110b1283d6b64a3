"""
Drop-in mirror of the reference's ``transform_dataloader.py``: camera poses and intrinsics from
COLMAP (``images/cameras`` ``.txt`` or ``.bin``, also under ``sparse/0``) or a ``transforms.json``.  Pure host parsing;
no OpenCV (image sizes missing from a transforms file are read from the PNG / JPEG header).  Results have the
reference's shape: ``name -> 4x4 camera-to-world (nested lists)``, ``name -> [w, h, fl_x, fl_y, ...]``.
Reference lines: transform_dataloader.py:8-299.
"""
import json
import os
import struct

import numpy as np

_FLIP = np.diag([1.0, -1.0, -1.0, 1.0])


def convert_sfm_pose_to_nerf(transform):
    """transform_dataloader.py:8-22 -- invert the world-to-camera pose, flip y and z (OpenCV -> OpenGL camera)."""
    return np.matmul(np.linalg.inv(transform), _FLIP)


def qvec2rotmat(qvec):
    """transform_dataloader.py:24-42."""
    w, x, y, z = qvec
    return np.array([[1 - 2 * y ** 2 - 2 * z ** 2, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x ** 2 - 2 * z ** 2, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x ** 2 - 2 * y ** 2]])


def read_next_bytes(fid, num_bytes, format_char_sequence, endian_character="<"):
    return struct.unpack(endian_character + format_char_sequence, fid.read(num_bytes))


def get_colmap_bin_intrinsics(file_path):
    """transform_dataloader.py:50-71 -- camera id -> (width, height, fx, fy, cx, cy)."""
    out = {}
    with open(file_path, "rb") as f:
        for _ in range(read_next_bytes(f, 8, "Q")[0]):
            elems = read_next_bytes(f, 56, "iiQQdddd")
            if elems[1] != 1:
                print("WARNING: Colmap cameras are a not Pinhole camera type. Rendered Colour quality might be impacted!")
            out[elems[0]] = elems[2:]
    return out


def get_colmap_txt_intrinsics(file_path):
    """transform_dataloader.py:73-96."""
    out = {}
    with open(file_path, "r") as f:
        for line in f:
            line = line.strip()
            if len(line) != 0 and line[0] == "#":
                continue
            elems = line.split(" ")
            if elems[1].lower().strip() != "pinhole":
                print("WARNING: Colmap cameras are not a Pinhole camera type. Rendered Colour quality might be impacted!")
            out[int(elems[0])] = elems[2:]
    return out


def get_colmap_img_transform(elems):
    """transform_dataloader.py:98-117 -- (id, qw qx qy qz, tx ty tz, ...) -> camera-to-world in the NeRF convention."""
    qvec = np.array(tuple(map(float, elems[1:5])))
    tvec = np.array(tuple(map(float, elems[5:8])))
    w2c = np.concatenate([np.concatenate([qvec2rotmat(-qvec), tvec.reshape([3, 1])], 1),
                          np.array([0.0, 0.0, 0.0, 1.0]).reshape([1, 4])], 0)
    return convert_sfm_pose_to_nerf(w2c).tolist()


def load_colmap_bin_data(input_path, skip_rate=0):
    """transform_dataloader.py:119-171."""
    transforms, cameras = {}, {}
    colmap_cameras = get_colmap_bin_intrinsics(os.path.join(input_path, "cameras.bin"))
    i = 0
    with open(os.path.join(input_path, "images.bin"), "rb") as f:
        for _ in range(read_next_bytes(f, 8, "Q")[0]):
            elems = read_next_bytes(f, 64, "idddddddi")
            transform = get_colmap_img_transform(elems)
            name = b""
            ch = f.read(1)
            while ch != b"\x00":
                name += ch
                ch = f.read(1)
            n2d = read_next_bytes(f, 8, "Q")[0]
            f.seek(24 * n2d, os.SEEK_CUR)
            if i % (skip_rate + 1) == 0:
                key = os.path.basename(name.decode("utf-8")).split('.')[0]
                transforms[key] = transform
                cameras[key] = colmap_cameras[elems[8]]
            i += 1
    return transforms, cameras


def load_colmap_txt_data(input_path, skip_rate=0):
    """transform_dataloader.py:173-211 (line-parity logic kept as in the reference)."""
    transforms, cameras = {}, {}
    colmap_cameras = get_colmap_txt_intrinsics(os.path.join(input_path, "cameras.txt"))
    i = 0
    with open(os.path.join(input_path, "images.txt"), "r") as f:
        for line in f:
            line = line.strip()
            if len(line) != 0 and line[0] == "#":
                continue
            i = i + 1
            if len(line) == 0:
                continue
            if i % 2 == 1 and i % (skip_rate + 1) == 0:
                elems = line.split(" ")
                key = os.path.basename(str(elems[9])).split('.')[0]
                transforms[key] = get_colmap_img_transform(elems)
                cameras[key] = colmap_cameras[int(elems[8])]
    return transforms, cameras


def image_size(fname):
    """(width, height) from a PNG or JPEG header (replaces cv2.imread(...).shape)."""
    with open(fname, "rb") as f:
        head = f.read(26)
        if head[:8] == b"\x89PNG\r\n\x1a\n":
            return struct.unpack(">II", head[16:24])
        if head[:2] == b"\xff\xd8":
            f.seek(2)
            while True:
                marker = f.read(2)
                while marker and marker[0:1] != b"\xff":
                    marker = marker[1:] + f.read(1)
                if len(marker) < 2:
                    break
                size = struct.unpack(">H", f.read(2))[0]
                if 0xC0 <= marker[1] <= 0xCF and marker[1] not in (0xC4, 0xC8, 0xCC):
                    h, w = struct.unpack(">xHH", f.read(5))
                    return w, h
                f.seek(size - 2, os.SEEK_CUR)
    raise Exception(f"Cannot read the size of image {fname} (PNG and JPEG headers are supported)")


def get_transform_intrinsics(transforms, fname):
    """transform_dataloader.py:213-247."""
    intrinsics = [0, 0, 0, 0]
    if "w" in transforms and "h" in transforms:
        intrinsics[0], intrinsics[1] = transforms["w"], transforms["h"]
    else:
        if not os.path.exists(fname):
            raise Exception(f"Image with path {fname} does not exist")
        intrinsics[0], intrinsics[1] = image_size(fname)
    if "fl_x" in transforms.keys():
        intrinsics[2] = transforms["fl_x"]
    elif "camera_angle_x" in transforms.keys():
        intrinsics[2] = 0.5 * intrinsics[0] / np.tan(0.5 * transforms["camera_angle_x"])
    else:
        raise Exception("A focal length (fl_x) or field of view (camera_angle_x) must be provided")
    if "fl_y" in transforms.keys():
        intrinsics[3] = transforms["fl_y"]
    elif "camera_angle_y" in transforms.keys():
        intrinsics[3] = 0.5 * intrinsics[1] / np.tan(0.5 * transforms["camera_angle_y"])
    else:
        intrinsics[3] = intrinsics[2]
    return intrinsics


def load_transform_json_data(input_path, skip_rate=0):
    """transform_dataloader.py:249-278."""
    with open(input_path, "r") as f:
        transforms = json.load(f)
    out, intrinsics = {}, {}
    shared = None
    if "fl_x" in transforms.keys() or "camera_angle_x" in transforms.keys():
        shared = get_transform_intrinsics(transforms, transforms["frames"][0]["file_path"])
    for i, frame in enumerate(transforms["frames"]):
        key = os.path.basename(str(os.path.basename(frame["file_path"]))).split('.')[0]
        intrinsics[key] = get_transform_intrinsics(frame, frame["file_path"]) if shared is None else shared
        if i % (skip_rate + 1) == 0:
            out[key] = frame["transform_matrix"]
    return out, intrinsics


def load_transform_data(input_path, skip_rate=0):
    """transform_dataloader.py:280-299."""
    if os.path.isdir(input_path):
        for base in (input_path, os.path.join(input_path, "sparse", "0")):
            if os.path.exists(os.path.join(base, "images.txt")):
                return load_colmap_txt_data(base, skip_rate=skip_rate)
            if os.path.exists(os.path.join(base, "images.bin")):
                return load_colmap_bin_data(base, skip_rate=skip_rate)
    elif os.path.splitext(input_path)[1] == ".json":
        return load_transform_json_data(input_path, skip_rate=skip_rate)
    raise AttributeError("Unsupported transform data type")
