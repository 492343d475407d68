"""Result wire format of the OpenLane-V evaluation: one `<ImgName>.lines.txt` per frame, one line per
predicted lane, "x y " pairs with one decimal.  Mirrors the reference's writer (evaluation/generate_lane.py:46-61, called from
testOLV3.py:110 for every frame): lanes with at most two points are dropped, points are written last-to-first, and a
normalised point (tx, ty) of the cropped, resized network input maps to the half-resolution original image as
x = tx * W_org / 2, y = (ty * H_crop + 480) / 2 (480 = rows cropped off the top, cfg.crop_size; size = (H_crop, W_org))."""
import os
from typing import Iterable, Sequence

CROP_TOP = 480     # evaluation/generate_lane.py:59 (OpenLane images 1280 x 1920, the top 480 rows are cropped before resizing)


def format_pred_lines(lanes: Iterable, size: Sequence[float]) -> str:
    """The text of one .lines.txt file.  lanes: objects with `.points` [n,2] normalised (x, y); size = (H_crop, W_org)."""
    out = []
    for lane in lanes:
        pts = lane.points
        if len(pts) > 2:
            out.append("".join("%.1f %.1f " % (tx * size[1] / 2, (ty * size[0] + CROP_TOP) / 2) for tx, ty in reversed(pts)) + "\n")
    return "".join(out)


def generate_predV2(info: dict, lanes: Iterable, img_num: int, pred_txt_path: str = "./evaluation/txt4OL/pred_txt") -> str:
    """Writes `<pred_txt_path>/<info['name']>/<info['ImgName'][img_num]>.lines.txt` and returns its path."""
    folder = os.path.join(pred_txt_path, info["name"])
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, info["ImgName"][img_num] + ".lines.txt")
    with open(path, "w") as fp:
        fp.write(format_pred_lines(lanes, info["size"]))
    return path
