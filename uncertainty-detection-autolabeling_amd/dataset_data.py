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

WAYMO_LABELS = {1: "vehicle", 2: "pedestrian", 3: "cyclist"}
VOC_LABELS = dict(enumerate(
    "aeroplane bicycle bird boat bottle bus car cat chair cow diningtable dog horse motorbike person pottedplant "
    "sheep sofa train tvmonitor".split(), start=1))
# COCO category ids are sparse (12, 26, 29, 30, 45, 66, 68, 69, 71, 83 are unused)
_COCO_NAMES = ("person,bicycle,car,motorcycle,airplane,bus,train,truck,boat,traffic light,fire hydrant,,stop sign,"
               "parking meter,bench,bird,cat,dog,horse,sheep,cow,elephant,bear,zebra,giraffe,,backpack,umbrella,,,handbag,"
               "tie,suitcase,frisbee,skis,snowboard,sports ball,kite,baseball bat,baseball glove,skateboard,surfboard,"
               "tennis racket,bottle,,wine glass,cup,fork,knife,spoon,bowl,banana,apple,sandwich,orange,broccoli,carrot,"
               "hot dog,pizza,donut,cake,chair,couch,potted plant,bed,,dining table,,,toilet,,tv,laptop,mouse,remote,"
               "keyboard,cell phone,microwave,oven,toaster,sink,refrigerator,,book,clock,vase,scissors,teddy bear,"
               "hair drier,toothbrush")
COCO_LABELS = {i: n for i, n in enumerate(_COCO_NAMES.split(","), start=1) if n}

_LABEL_MAPS = {"kitti": KITTI_LABELS, "bdd": BDD_LABELS, "coco": COCO_LABELS, "voc": VOC_LABELS, "waymo": WAYMO_LABELS}


def get_label_map(mapping):
    """name | yaml path | dict | Config | None -> {class id: name}  (reference label_util.get_label_map, :170-188)."""
    if not mapping:
        return None
    if isinstance(mapping, dict):
        return {int(k): v for k, v in mapping.items()}
    if hasattr(mapping, "as_dict"):
        return {int(k): v for k, v in mapping.as_dict().items()}
    if not isinstance(mapping, str):
        raise ValueError("mapping must be dict or str.")
    if mapping.endswith(".yaml"):
        import yaml
        with open(mapping) as f:
            return yaml.load(f, Loader=yaml.FullLoader)
    if mapping in _LABEL_MAPS:
        return dict(_LABEL_MAPS[mapping])
    raise KeyError(mapping)


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
