"""Host-side drawing behind `ServingDriver.visualize` (reference src/infer_lib.py:46-91,194-204).

The reference delegates to the TF object-detection `vis_utils` (2.8 kLoC of drawing utilities, out of scope); this
is a small PIL drawer with the same call shape and the same selection rules (score threshold, at most
`max_boxes_to_draw` boxes in the given order, label "name: score%"), so `driver.visualize(...)` keeps working for
the callers (infer_model.py:787+).  Not on the hot path: <= 100 boxes per image.
"""
import numpy as np

from .dataset_data import get_label_map

_COLORS = [(230, 25, 75), (60, 180, 75), (255, 225, 25), (0, 130, 200), (245, 130, 48), (145, 30, 180), (70, 240, 240),
           (240, 50, 230), (210, 245, 60), (250, 190, 190), (0, 128, 128), (230, 190, 255), (170, 110, 40)]


def visualize_image(image, boxes, classes, scores, label_map=None, uncertainty=None, min_score_thresh=0.01,
                    max_boxes_to_draw=1000, line_thickness=2, **kwargs):
    """image [H,W,3] uint8; boxes [N,4] (ymin, xmin, ymax, xmax) in pixels; classes [N] int; scores [N];
    uncertainty [N,4] box std in pixels or None (drawn as a second, dashed-looking outline at +-1 std).
    Returns the annotated image as a uint8 array."""
    from PIL import Image, ImageDraw
    label_map = get_label_map(label_map or "coco") or {}
    img = Image.fromarray(np.asarray(image).astype(np.uint8)).convert("RGB")
    draw = ImageDraw.Draw(img)
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    classes = np.asarray(classes).reshape(-1).astype(int)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    unc = None if uncertainty is None else np.asarray(uncertainty, dtype=np.float32).reshape(len(boxes), -1)
    drawn = 0
    for i in range(len(boxes)):
        if drawn >= max_boxes_to_draw:
            break
        if scores[i] < min_score_thresh:
            continue
        ymin, xmin, ymax, xmax = [float(v) for v in boxes[i]]
        if not (ymax > ymin and xmax > xmin):
            continue
        color = _COLORS[classes[i] % len(_COLORS)]
        draw.rectangle([xmin, ymin, xmax, ymax], outline=color, width=int(line_thickness))
        if unc is not None and unc.shape[1] >= 4 and np.all(np.isfinite(unc[i, :4])):
            sy0, sx0, sy1, sx1 = [float(v) for v in unc[i, :4]]
            draw.rectangle([xmin - sx0, ymin - sy0, xmax + sx1, ymax + sy1], outline=color, width=1)
            if xmax - sx1 > xmin + sx0 and ymax - sy1 > ymin + sy0:
                draw.rectangle([xmin + sx0, ymin + sy0, xmax - sx1, ymax - sy1], outline=color, width=1)
        name = label_map.get(int(classes[i]), str(int(classes[i])))
        text = "%s: %d%%" % (name, int(round(100 * float(scores[i]))))
        tw = 6 * len(text) + 2
        ty = ymin - 11 if ymin >= 11 else ymin
        draw.rectangle([xmin, ty, xmin + tw, ty + 11], fill=color)
        draw.text((xmin + 1, ty), text, fill=(0, 0, 0))
        drawn += 1
    return np.asarray(img)
