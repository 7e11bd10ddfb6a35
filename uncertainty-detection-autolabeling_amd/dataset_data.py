"""Dataset registry for the path: label map, class names and raw image shape.

Mirrors the part of the reference's ``src/dataset_data.py:85-130``
(`get_dataset_data`) and ``src/label_util.py:144-167`` that the serving path
and its synthetic benchmark shapes need.  Label files, occlusion/truncation
parsing and image folders are dataset plumbing outside the hot path.
"""

KITTI_LABELS = {1: "car", 2: "van", 3: "truck", 4: "pedestrian",
                5: "person_sitting", 6: "cyclist", 7: "tram"}
BDD_LABELS = {1: "pedestrian", 2: "rider", 3: "car", 4: "truck", 5: "bus", 6: "train",
              7: "motorcycle", 8: "bicycle", 9: "traffic light", 10: "traffic sign"}

_LABEL_MAPS = {"kitti": KITTI_LABELS, "bdd": BDD_LABELS}


def get_label_map(mapping):
    """'kitti' | 'bdd' | dict | None -> {class id: name}."""
    if not mapping:
        return None
    if isinstance(mapping, dict):
        return {int(k): v for k, v in mapping.items()}
    if mapping in _LABEL_MAPS:
        return dict(_LABEL_MAPS[mapping])
    raise ValueError("unknown label map %r" % (mapping,))


def get_dataset_data(path, im_name=None):
    """(label_map, img_source_path, class_names, raw [H, W], img_file) keyed on the
    dataset name contained in `path` (dataset_data.py:95-128)."""
    label_map, img_source_path, img_shape, class_names = {}, None, [0, 0], []
    if "KITTI" in path:
        label_map = get_label_map("kitti")
        img_source_path = "/KITTI/training/image_2/"
        img_shape = [375, 1220]
    elif "BDD" in path:
        label_map = get_label_map("bdd")
        img_source_path = "/BDD100K/bdd100k/images/100k/val/"
        img_shape = [720, 1280]
    elif "CODA" in path:
        label_map = get_label_map("bdd")
        img_source_path = "/CODA/images/"
        img_shape = [1000, 1500]
    if label_map and "CODA" not in path:
        class_names = [label_map[i].capitalize() for i in range(1, len(label_map) + 1)]
    img_file = (img_source_path + im_name) if (im_name and img_source_path) else None
    return label_map, img_source_path, class_names, img_shape, img_file
