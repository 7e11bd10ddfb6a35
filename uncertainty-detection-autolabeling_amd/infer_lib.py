"""ServingDriver-shaped boundary over the HIP C-ABI library.

Drop-in for the reference's `infer_lib` driver classes (src/infer_lib.py:118-491) as their
callers use them (inspector.py:161-169; validate_model.py:155; infer_model.py:581,787;
calibrate_model.py:89; utils_extra.py:130-133) - same class names, same positional
arguments, same members:

    driver = ServingDriver.create(model_dir, debug, saved_model_dir, model_name,
                                  batch_size, only_network, model_params)       # :154-163
    driver = KerasDriver(ckpt_path, debug, model_name, batch_size, only_network, model_params)        # :416-440
    driver = SavedModelDriver(saved_model_dir, model_name, batch_size, only_network, model_params)    # :299-311
    boxes, scores, classes, valid_len[, logits] = driver.serve(uint8_images)
    cls_outputs, box_outputs = driver.predict(float_images)      # only_network=True
    driver.benchmark(images, bm_runs=10)
    driver.visualize(image, boxes, classes, scores, uncertainty=None)

Output layout = `postprocess_global` (src/postprocess.py:610-621):
boxes [N,100,4(+4 aleatoric)(+4 epistemic)], scores [N,100], classes [N,100] or
[N,100,1+C], valid_len [N] int32, logits [N,100,C] when `enable_softmax`.
Arrays are fresh numpy arrays owned by the caller; errors are Python exceptions.

Weights: there is no TF runtime here, so the path argument (`ckpt_path` / `saved_model_dir`)
names an `.npz` weight set with the reference's variable names (weights.py), or a TF2
checkpoint prefix / directory read by `ckpt_reader` (no TensorFlow needed); "_" (the
reference's "running test: do not load any ckpt", utils_keras.py:142-144) or an empty path
draws a random-init set.  Keyword-only extras (device, chunk_images, weights, post_mode) are
the build's additions and never positional, so the reference's call expressions bind unchanged.
"""
import ctypes as C
import logging
import time

import numpy as np

from . import capi, hparams_config, plan as plan_mod, weights as weights_mod


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class ServingDriver:
    """One GPU, one handle, synchronous calls (same threading contract as the reference).

    `__init__(model_name, batch_size=1, only_network=False, model_params=None)` is the reference's
    base-class signature (infer_lib.py:165-192); callers construct `KerasDriver` / `SavedModelDriver`
    or go through `create`."""

    @classmethod
    def create(cls, model_dir, debug, saved_model_dir, *args, **kwargs):
        """infer_lib.py:154-163: a saved-model path selects SavedModelDriver (".tflite": TfliteDriver), else KerasDriver."""
        if saved_model_dir:
            if str(saved_model_dir).endswith("tflite"):
                return TfliteDriver(saved_model_dir, *args, **kwargs)
            return SavedModelDriver(saved_model_dir, *args, **kwargs)
        return KerasDriver(model_dir, debug, *args, **kwargs)

    def __init__(self, model_name, batch_size=1, only_network=False, model_params=None, *, weights_path=None,
                 weights=None, device=0, chunk_images=None, post_mode="global", debug=False, post_only=False):
        self.model_name = model_name
        self.batch_size = batch_size
        self.only_network = only_network
        self.debug = debug
        self.params = hparams_config.get_detection_config(model_name).as_dict()
        if model_params:
            self.params.update(model_params)          # a plain dict update, as the reference does (:186-187)
        self.params.update(dict(is_training_bn=False))
        self.label_map = self.params.get("label_map", None)
        if self.params["nms_configs"].get("pyfunc", False):
            raise ValueError("nms_configs.pyfunc=True selects the numpy NMS path (nms_np), "
                             "which is dead in the reference (postprocess.py:806) and not served here")
        self._cap = int(batch_size or 1)              # images one uda_run holds (batch_size None/0 = the reference's dynamic batch)

        if weights is None and not post_only:
            weights = weights_mod.resolve_weights(weights_path, self.params)
        self.weights = weights
        if chunk_images is None:
            chunk_images = min(self._cap, int(self.params.get("uda_chunk_images", 16)))
        # `uda_pw_scheme` in model_params (f16x2 | bf16x3 | bf16x2 | f32): the split scheme of THIS handle's 1x1 contractions.  The
        # planner and uda_create both read UDA_PW_SCHEME when they run; the key sets it for exactly that long (a
        # per-handle choice instead of a process-wide environment variable).
        import os
        want = self.params.get("uda_pw_scheme")
        saved = (os.environ.get("UDA_PW_SCHEME"), os.environ.get("UDA_PW_TERMS"))
        if want:
            if want not in plan_mod.PW_SCHEMES:
                raise ValueError("uda_pw_scheme=%r: expected one of %s" % (want, ", ".join(plan_mod.PW_SCHEMES)))
            os.environ["UDA_PW_SCHEME"] = want
            os.environ.pop("UDA_PW_TERMS", None)
        try:
            self.pw_scheme = plan_mod.pw_scheme()
            self._create(weights, chunk_images, post_only, post_mode, device)
        finally:
            if want:
                for k, v in zip(("UDA_PW_SCHEME", "UDA_PW_TERMS"), saved):
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        self.device = int(device)
        self.image_size = hparams_config.parse_image_size(self.params["image_size"])
        self.T = self.plan.T
        self.M = int(self.params["nms_configs"]["max_output_size"])
        self.num_classes = int(self.params["num_classes"])
        self._seed_counter = int(self.params.get("uda_dropout_seed", 0))
        self._fixed_seed = None
        self._run_id = 0                              # bumped by every call that rewrites the resident head outputs

    def _create(self, weights, chunk_images, post_only, post_mode, device):
        self.plan = plan_mod.Plan(self.params, weights, chunk_images=chunk_images, max_images=self._cap, post_only=post_only)
        if self.plan.unknown_keys:
            # not one of hparams_config's keys and not a `uda_*` knob: nothing here reads it (plan.MODEL_PARAM_HANDLING)
            logging.warning("model_params keys the HIP path does not know (ignored): %s", ", ".join(self.plan.unknown_keys))
        self._post_mode = capi.POST_PER_CLASS if post_mode == "per_class" else capi.POST_GLOBAL
        self._lib = capi.load()
        m, bufs, ops, sites, blob, anchors = self.plan.to_c(self._post_mode)
        self._keep = (m, bufs, ops, sites, blob, anchors)
        handle = C.c_void_p()
        rc = self._lib.uda_create(C.byref(m), bufs, len(bufs), ops, len(self.plan.ops), sites, _ptr(blob), blob.size,
                                  _ptr(anchors), int(device), C.byref(handle))
        if rc != 0:
            raise capi.UdaError("uda_create failed: %s" % self._lib.uda_last_error(None).decode())
        self._h = handle

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.uda_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        capi.check(self._lib, self._h, rc, what)

    # ------------------------------------------------------------------ MC dropout control
    def set_dropout_seed(self, seed):
        """Fix the Philox seed (every call then draws the same masks); None = advance per call."""
        self._fixed_seed = None if seed is None else int(seed)

    def set_dropout_masks(self, masks):
        """Inject masks: {site name: float32 [N, T, C]} (see plan.Plan.sites for the names)."""
        parts = []
        for name, ch, _ in self.plan.sites:
            a = np.ascontiguousarray(masks[name], dtype=np.float32)
            parts.append(a.reshape(-1, ch).reshape(-1))
        flat = np.concatenate(parts) if parts else np.zeros(0, np.float32)
        self._ck(self._lib.uda_set_dropout_masks(self._h, _ptr(flat), flat.size), "uda_set_dropout_masks")
        self._injected = True

    def _next_seed(self):
        if getattr(self, "_injected", False):
            return
        if self._fixed_seed is not None:
            seed = self._fixed_seed
        else:
            seed = self._seed_counter
            self._seed_counter += 1
        self._ck(self._lib.uda_set_dropout_seed(self._h, C.c_uint64(seed)), "uda_set_dropout_seed")

    # ------------------------------------------------------------------ serve / predict
    def _as_u8_batch(self, image_arrays):
        """uint8 [N,h,w,3] (an array, or a list of equally sized images as the reference stacks them, infer_lib.py:139-151)."""
        if isinstance(image_arrays, (list, tuple)):
            image_arrays = np.stack([np.asarray(a) for a in image_arrays])
        a = np.asarray(image_arrays)
        if a.ndim == 3:
            a = a[None]
        if a.ndim != 4 or a.shape[-1] != 3:
            raise ValueError("images must be [N, h, w, 3], got %s" % (a.shape,))
        if a.dtype != np.uint8:
            raise ValueError("serve() takes uint8 images, got %s" % a.dtype)
        if a.shape[0] > self._cap:
            raise ValueError("batch of %d images exceeds batch_size=%d" % (a.shape[0], self._cap))
        return np.ascontiguousarray(a)

    def _feed(self, image_arrays, prefetch=False):
        """Hand a uint8 batch to the handle: one array [N,h,w,3], or a list of images whose raw sizes may differ (KITTI:
        370-376 x 1224-1242, dataset_data.py:105 - the reference serves those one file at a time,
        validate_model.py:479-483; here they form one batch, each image with its own resize scale).  prefetch=True uploads
        into the second input slot on the copy stream (`swap_prefetched` makes it current).  Returns the image count."""
        ragged = (isinstance(image_arrays, (list, tuple)) and len(image_arrays) > 0
                  and len({np.shape(x) for x in image_arrays}) > 1)
        if not ragged:
            a = self._as_u8_batch(image_arrays)
            n, h, w = a.shape[:3]
            fn = self._lib.uda_prefetch_images_u8 if prefetch else self._lib.uda_set_images_u8
            self._ck(fn(self._h, _ptr(a), n, h, w), "uda_prefetch_images_u8" if prefetch else "uda_set_images_u8")
            return n
        imgs = []
        for x in image_arrays:
            x = np.asarray(x)
            if x.ndim != 3 or x.shape[-1] != 3:
                raise ValueError("every image of a ragged batch must be [h, w, 3], got %s" % (x.shape,))
            if x.dtype != np.uint8:
                raise ValueError("serve() takes uint8 images, got %s" % x.dtype)
            imgs.append(np.ascontiguousarray(x))
        n = len(imgs)
        if n > self._cap:
            raise ValueError("batch of %d images exceeds batch_size=%d" % (n, self._cap))
        ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in imgs])
        hs = np.asarray([x.shape[0] for x in imgs], np.int32)
        ws = np.asarray([x.shape[1] for x in imgs], np.int32)
        fn = self._lib.uda_prefetch_images_u8_ragged if prefetch else self._lib.uda_set_images_u8_ragged
        self._ck(fn(self._h, ptrs, n, _ptr(hs), _ptr(ws)), "uda_set_images_u8_ragged")
        return n

    def _mode(self, post_mode):
        if post_mode is None:
            return self._post_mode
        if post_mode in ("global", capi.POST_GLOBAL):
            return capi.POST_GLOBAL
        if post_mode in ("per_class", capi.POST_PER_CLASS):
            return capi.POST_PER_CLASS
        raise ValueError("Unsupported postprocess mode {}".format(post_mode))

    def _collect(self, n, mode=None):
        mode = self._post_mode if mode is None else mode
        bc, cc = C.c_int32(), C.c_int32()
        self._ck(self._lib.uda_detection_cols(self._h, mode, C.byref(bc), C.byref(cc)), "uda_detection_cols")
        boxes = np.empty((n, self.M, bc.value), np.float32)
        scores = np.empty((n, self.M), np.float32)
        classes = np.empty((n, self.M, cc.value), np.float32)
        valid = np.empty((n,), np.int32)
        with_logits = self.params["enable_softmax"] and mode == capi.POST_GLOBAL
        logits = np.empty((n, self.M, self.num_classes), np.float32) if with_logits else None
        self._ck(self._lib.uda_get_detections(self._h, _ptr(boxes), _ptr(scores), _ptr(classes), _ptr(valid),
                                              _ptr(logits)), "uda_get_detections")
        if cc.value == 1:
            classes = classes[..., 0]
        out = [boxes, scores, classes, valid]
        if logits is not None:
            out.append(logits)
        return tuple(out)

    def run_async(self, post_mode=None):
        """Queue network + post-process of the batch that is set (`stage_images`, `prefetch_images` + `swap_prefetched`)
        WITHOUT ordering the handle behind the post-process: the next `run_async` starts its network at once, the ~4 ms of
        aggregate / NMS / gather launches of this batch run beside it (`uda_run_async`).  Returns a ticket for `collect`;
        at most two runs may be in flight."""
        mode = self._mode(post_mode)
        self._next_seed()
        self._run_id += 1
        t = C.c_int32(-1)
        self._ck(self._lib.uda_run_async(self._h, mode, C.byref(t)), "uda_run_async")
        if not hasattr(self, "_tickets"):
            self._tickets = {}
        self._tickets[t.value] = (self._n_last(), mode)
        return t.value

    def drain(self):
        """Abandon every pipelined run in flight (`uda_drain`): results discarded, tickets closed."""
        if getattr(self, "_h", None):
            self._ck(self._lib.uda_drain(self._h), "uda_drain")
        if hasattr(self, "_tickets"):
            self._tickets.clear()

    def range_demotions(self):
        """Ops re-packed with three bf16 pieces because an operand exceeded fp16's range (`uda_range_demotions`)."""
        return int(self._lib.uda_range_demotions(self._h))

    def collect(self, ticket):
        """The detections of a `run_async` (same tuple as `serve`): waits for that run's post-process only."""
        n, mode = self._tickets.pop(ticket)
        bc, cc = C.c_int32(), C.c_int32()
        self._ck(self._lib.uda_detection_cols(self._h, mode, C.byref(bc), C.byref(cc)), "uda_detection_cols")
        boxes = np.empty((n, self.M, bc.value), np.float32)
        scores = np.empty((n, self.M), np.float32)
        classes = np.empty((n, self.M, cc.value), np.float32)
        valid = np.empty((n,), np.int32)
        with_logits = self.params["enable_softmax"] and mode == capi.POST_GLOBAL
        logits = np.empty((n, self.M, self.num_classes), np.float32) if with_logits else None
        self._ck(self._lib.uda_collect(self._h, ticket, _ptr(boxes), _ptr(scores), _ptr(classes), _ptr(valid), _ptr(logits)),
                 "uda_collect")
        if cc.value == 1:
            classes = classes[..., 0]
        out = [boxes, scores, classes, valid]
        if logits is not None:
            out.append(logits)
        return tuple(out)

    def collect_device(self, ticket, rows=None):
        """`collect` for the multi-GPU gather: the run's detections as one device-resident record buffer
        (device address, rows, layout) - see `detections_device`; closes the ticket."""
        n, mode = self._tickets.pop(ticket)
        rows = n if rows is None else int(rows)
        bc, cc = C.c_int32(), C.c_int32()
        self._ck(self._lib.uda_detection_cols(self._h, mode, C.byref(bc), C.byref(cc)), "uda_detection_cols")
        with_logits = bool(self.params["enable_softmax"] and mode == capi.POST_GLOBAL)
        ptr, cols = C.c_void_p(), C.c_int32()
        self._ck(self._lib.uda_collect_device(self._h, ticket, rows, int(with_logits), C.byref(ptr), C.byref(cols)), "uda_collect_device")
        layout = dict(box=bc.value, cls=cc.value, logits=self.num_classes if with_logits else 0)
        assert cols.value == layout["box"] + 1 + layout["cls"] + layout["logits"] + 1
        return ptr.value, rows, layout

    def detections_device(self, rows=None, mode=None):
        """The detections of the last run as ONE device-resident record buffer: (device address, rows, layout) with the
        float32 buffer [rows, M, cols] laid out as `dist.pack_detections` does on the host.  rows >= the images of the last
        run pads the tail with zeros (ragged shards); `dist.all_gather_detections_device` gathers straight out of it."""
        mode = self._post_mode if mode is None else mode
        n = self._n_last()
        rows = n if rows is None else int(rows)
        bc, cc = C.c_int32(), C.c_int32()
        self._ck(self._lib.uda_detection_cols(self._h, mode, C.byref(bc), C.byref(cc)), "uda_detection_cols")
        with_logits = bool(self.params["enable_softmax"] and mode == capi.POST_GLOBAL)
        ptr, cols = C.c_void_p(), C.c_int32()
        self._ck(self._lib.uda_detections_device(self._h, rows, int(with_logits), C.byref(ptr), C.byref(cols)), "uda_detections_device")
        layout = dict(box=bc.value, cls=cc.value, logits=self.num_classes if with_logits else 0)
        assert cols.value == layout["box"] + 1 + layout["cls"] + layout["logits"] + 1
        return ptr.value, rows, layout

    def empty_detections(self):
        """A zero-image output tuple with the right trailing shapes (ragged multi-GPU shards)."""
        bc, cc = C.c_int32(), C.c_int32()
        self._ck(self._lib.uda_detection_cols(self._h, -1, C.byref(bc), C.byref(cc)), "uda_detection_cols")
        cls_shape = (0, self.M) if cc.value == 1 else (0, self.M, cc.value)
        out = [np.zeros((0, self.M, bc.value), np.float32), np.zeros((0, self.M), np.float32),
               np.zeros(cls_shape, np.float32), np.zeros((0,), np.int32)]
        if self.params["enable_softmax"]:
            out.append(np.zeros((0, self.M, self.num_classes), np.float32))
        return tuple(out)

    def set_image_offset(self, first_image):
        """Position of this driver's first image in the global batch (multi-GPU shards): makes the
        Philox dropout rows those of the unsharded batch."""
        self._ck(self._lib.uda_set_dropout_image_offset(self._h, int(first_image)), "uda_set_dropout_image_offset")

    def set_sample_shard(self, first, stride, total):
        """This driver's T samples are samples first + j * stride of a global axis of `total` MC samples (sample sharding over
        ranks, `dist.serve_sample_sharded`): its dropout rows are those of one driver that runs all of them.  total = 0: off."""
        self._ck(self._lib.uda_set_dropout_sample_shard(self._h, int(first), int(stride), int(total)), "uda_set_dropout_sample_shard")

    def run_network(self, image_arrays):
        """uint8 images -> preprocess + network for this driver's samples, no post-process: the head outputs stay in the handle
        (`head_outputs`, `head_outputs_device`).  Returns the image count."""
        n = self._feed(image_arrays)
        self._next_seed()
        self._run_id += 1
        self._ck(self._lib.uda_run(self._h, -1, 0), "uda_run")
        self._last_n = n
        return n

    def serve(self, image_arrays, post_mode=None):
        """uint8 [N,h,w,3] -> (boxes, scores, classes, valid_len[, logits]).

        post_mode None = the driver's default ("global", as `EfficientDetModel.call`); "per_class"
        gives `postprocess_per_class` (boxes [N,M,4], scores, classes [N,M], valid_len): the
        uncertainty columns are dropped there by the reference (postprocess.py:737) and its logits
        output in that mode is corrupted by a variable overwrite (:659-666), so none is returned."""
        mode = self._mode(post_mode)
        n = self._feed(image_arrays)
        self._next_seed()
        self._run_id += 1
        self._ck(self._lib.uda_run(self._h, mode, 1), "uda_run")
        self._last_n = n
        return self._collect(n, mode)

    def serve_resident(self, image_arrays, post_mode=None):
        """serve() without the download: the detections stay in the handle (`detections_device`, calibrators,
        `class_probs`) - what the multi-GPU layer runs before its device-resident gather.  Returns the image count."""
        mode = self._mode(post_mode)
        n = self._feed(image_arrays)
        self._next_seed()
        self._run_id += 1
        self._ck(self._lib.uda_run(self._h, mode, 1), "uda_run")
        self._last_n = n
        return n

    def serve_files(self, paths, post_mode=None):
        """Decode image files and serve them as ONE batch (the reference reads and serves file by file,
        validate_model.py:479-483, infer_model.py:554-581); raw sizes may differ."""
        return self.serve(read_images(paths), post_mode=post_mode)

    def serve_stream(self, batches, post_mode=None, while_resident=None):
        """Generator over batches (arrays or lists of images): yields each batch's detections, with the upload of batch
        i + 1 (pinned staging buffer, copy stream) running under the network of batch i - the feed the reference pays
        inside every serve() call (validate_model.py:154-158) costs no device time here.
        while_resident(detections): optional callable run while the batch's outputs are still resident in the handle
        (softmax / entropy, calibrators: everything `class_probs`, `BoxCalibrator`, `ClassCalibrator` read); its return
        value is yielded instead of the detection tuple."""
        mode = self._mode(post_mode)
        it = iter(batches)
        try:
            first = next(it)
        except StopIteration:
            return
        n = self._feed(first)
        if while_resident is None:
            # nothing has to look at a batch's outputs inside the handle: the batches are pipelined - batch i + 1's network is
            # queued before batch i's detections are fetched, batch i's post-process runs beside it (uda_run_async)
            self._last_n = n
            t = self.run_async(mode)
            try:
                while True:
                    nxt = next(it, None)
                    if nxt is not None:
                        n_next = self._feed(nxt, prefetch=True)
                        self._ck(self._lib.uda_swap_prefetched(self._h), "uda_swap_prefetched")
                        self._last_n = n_next
                        t_next = self.run_async(mode)
                    yield self.collect(t)
                    if nxt is None:
                        return
                    t = t_next
            finally:
                # the consumer dropped the generator, or a collect / a feed raised: a run may still be queued - abandon it, so
                # that the handle serves synchronously again (every such entry point refuses beside a run in flight)
                self.drain()
        while True:
            self._next_seed()
            self._run_id += 1
            self._ck(self._lib.uda_run(self._h, mode, 1), "uda_run")          # queued, asynchronous
            nxt = next(it, None)
            n_next = self._feed(nxt, prefetch=True) if nxt is not None else 0  # overlaps the kernels queued above
            self._last_n = n
            det = self._collect(n, mode)                                      # waits for this batch
            yield det if while_resident is None else while_resident(det)
            if nxt is None:
                return
            self._ck(self._lib.uda_swap_prefetched(self._h), "uda_swap_prefetched")
            n = n_next

    def class_probs(self, n):
        """(probab [n, M, C], entropy [n, M]) of the last global post-process, computed on the device:
        `stable_softmax(logits)` and `-sum p * log2(max(p, 1e-7))` exactly as every caller of `serve` does next
        (validate_model.py:159-166, infer_model.py:585-600, utils_class.py:36-41)."""
        probs = np.empty((n, self.M, self.num_classes), np.float32)
        ent = np.empty((n, self.M), np.float32)
        self._ck(self._lib.uda_get_class_probs(self._h, _ptr(probs), _ptr(ent)), "uda_get_class_probs")
        return probs, ent

    def serve_unpacked(self, image_arrays):
        """serve + the unpacking `Validate._process_val_image` / `Infer` do on the host (validate_model.py:159-202):
        dict(boxes [N,M,4], scores, classes [N,M], valid_len, logits, probab, entropy, albox, mcbox, mcclass);
        entries a configuration does not produce are None."""
        from . import postprocess as pp
        det = self.serve(image_arrays)
        n = det[0].shape[0]
        probs = ent = None
        if self.params["enable_softmax"]:
            probs, ent = self.class_probs(n)
        return pp.unpack_detections(self.params, det, probs, ent)

    def predict_resident(self, image_arrays):
        """The raw network on float32 [N,H,W,3] inputs; the head outputs stay in the handle (`device_heads`)."""
        a = np.ascontiguousarray(image_arrays, dtype=np.float32)
        H, W = self.image_size
        if a.ndim != 4 or a.shape[1:] != (H, W, 3):
            raise ValueError("predict() takes float images [N,%d,%d,3], got %s" % (H, W, a.shape))
        if a.shape[0] > self._cap:
            raise ValueError("batch of %d images exceeds batch_size=%d" % (a.shape[0], self._cap))
        self._next_seed()
        self._run_id += 1
        self._ck(self._lib.uda_predict(self._h, _ptr(a), a.shape[0]), "uda_predict")
        self._last_n = a.shape[0]
        return a.shape[0]

    def predict(self, image_arrays):
        """only_network: float32 [N,H,W,3] -> (cls_outputs[levels], box_outputs[levels]) with the
        reference's shapes ([N,h,w,ch], or [T,N,h,w,ch] for a head that is MC-stacked);
        otherwise identical to serve()."""
        if not self.only_network:
            return self.serve(image_arrays)
        return self.head_outputs(self.predict_resident(image_arrays))

    def head_outputs(self, n):
        """Raw head outputs of the last run in the reference's layout: [N,h,w,ch], or [T,N,h,w,ch] for a head the
        reference stacks (its own or the global dropout rate is non-zero, efficientdet_keras.py:1026-1049).  A head
        the reference stacks but whose T samples are identical here (all its upstream rates are zero: one copy on
        the device) is broadcast to the stacked shape."""
        p = self.plan
        A = len(self.params["aspect_ratios"]) * self.params["num_scales"]
        cls_ch = A * self.num_classes
        box_ch = A * (8 if self.params["loss_attenuation"] else 4)
        cls_out, box_out = [], []
        for lvl, (h, w) in enumerate(p.level_hw):
            tc = self.T if p.cls_stacked_dev else 1
            tb = self.T if p.box_stacked_dev else 1
            c = np.empty((tc, n, h, w, cls_ch), np.float32)
            b = np.empty((tb, n, h, w, box_ch), np.float32)
            self._ck(self._lib.uda_get_head_outputs(self._h, lvl, _ptr(c), _ptr(b)), "uda_get_head_outputs")
            for out, arr, stacked, t_dev in ((cls_out, c, p.cls_stacked, tc), (box_out, b, p.box_stacked, tb)):
                if stacked and self.T > 1:
                    out.append(arr if t_dev == self.T else np.broadcast_to(arr, (self.T,) + arr.shape[1:]))
                else:
                    out.append(arr[0])
        return cls_out, box_out

    def _head_array(self, arr, stacked_dev, lvl, ch, what):
        """One level of injected head outputs -> contiguous float32 [T_dev, n, h, w, ch]; raises on any other shape
        (the C side reads exactly T_dev * n * h * w * ch floats)."""
        h, w = self.plan.level_hw[lvl]
        a = np.asarray(arr, dtype=np.float32)
        t_dev = self.T if stacked_dev else 1
        if a.ndim == 4:
            a = a[None]
        if a.ndim != 5 or a.shape[2:] != (h, w, ch):
            raise ValueError("%s outputs of level %d must be [(T,) N, %d, %d, %d], got %s" % (what, lvl, h, w, ch, np.shape(arr)))
        if a.shape[0] != t_dev:
            if t_dev == 1 and a.shape[0] == self.T:
                # stacked by the reference's rules, one copy on the device (identical samples): they must agree
                if not all(np.array_equal(a[0], a[t]) for t in range(1, a.shape[0])):
                    raise ValueError("%s outputs of level %d carry %d different samples, but this configuration has no "
                                     "dropout upstream of that head" % (what, lvl, a.shape[0]))
                a = a[:1]
            else:
                raise ValueError("%s outputs of level %d carry %d samples on axis 0, the handle expects %d"
                                 % (what, lvl, a.shape[0], t_dev))
        return np.ascontiguousarray(a)

    def postprocess(self, cls_outputs, box_outputs, image_scales=None, post_mode=None, collect=True):
        """`ServingDriver._postprocess` = postprocess_global on given head outputs (infer_lib.py:263-267);
        post_mode="per_class" = postprocess_per_class (eval.py:117-123 via generate_detections).  Head outputs that
        are still resident in this handle (`DeviceHeads` of its last run) are not uploaded again."""
        mode = self._mode(post_mode)
        p = self.plan
        resident = (isinstance(cls_outputs, DeviceHeads) and isinstance(box_outputs, DeviceHeads)
                    and cls_outputs.driver is self and cls_outputs.run_id == self._run_id == box_outputs.run_id)
        if resident:
            n = cls_outputs.n
        else:
            if len(cls_outputs) != len(p.level_hw) or len(box_outputs) != len(p.level_hw):
                raise ValueError("expected %d pyramid levels of head outputs" % len(p.level_hw))
            A = len(self.params["aspect_ratios"]) * self.params["num_scales"]
            cls_ch, box_ch = A * self.num_classes, A * (8 if self.params["loss_attenuation"] else 4)
            cs = [self._head_array(cls_outputs[l], p.cls_stacked_dev, l, cls_ch, "class") for l in range(len(p.level_hw))]
            bs = [self._head_array(box_outputs[l], p.box_stacked_dev, l, box_ch, "box") for l in range(len(p.level_hw))]
            n = cs[0].shape[1]
            if n > self._cap or any(x.shape[1] != n for x in cs + bs):
                raise ValueError("head outputs hold %s images, the handle at most %d" % (sorted({x.shape[1] for x in cs + bs}), self._cap))
            self._run_id += 1
            for lvl, (c, b) in enumerate(zip(cs, bs)):
                self._ck(self._lib.uda_set_head_outputs(self._h, lvl, n, _ptr(c), c.size, _ptr(b), b.size), "uda_set_head_outputs")
        if self.params.get("uncert_adjust_method") == "sample" and self.params.get("loss_attenuation"):
            self._next_seed()         # the "sample" decode draws from the handle's Philox stream (seed as for MC dropout)
        s = None if image_scales is None else np.ascontiguousarray(image_scales, dtype=np.float32)
        if s is not None and s.shape != (n,):
            raise ValueError("image_scales must have shape (%d,), got %s" % (n, s.shape))
        self._ck(self._lib.uda_postprocess_heads(self._h, n, _ptr(s), mode), "uda_postprocess_heads")
        self._last_n = n
        return self._collect(n, mode) if collect else None      # (collect=False: the detections stay resident, see detections_device)

    def device_heads(self, n):
        """(cls_outputs, box_outputs) of the last run as lazy sequences that stay on the device until indexed."""
        return DeviceHeads(self, n, 0), DeviceHeads(self, n, 1)

    def heads_written_externally(self, n):
        """Another producer (the ensemble exchange: device-to-device copies / RCCL receives into `head_outputs_device`
        buffers) has filled the head outputs of `n` images: earlier `DeviceHeads` views become stale, the next
        post-process works on n images."""
        self._run_id += 1
        if n > 0:
            self._ck(self._lib.uda_set_num_images(self._h, int(n)), "uda_set_num_images")
        self._last_n = int(n)

    def head_outputs_device(self, level, which):
        """(device address, floats per row, rows per image) of the handle's head-output buffer (capi.DevArray wraps it
        for torch / RCCL): rows are [image][sample]."""
        ptr, fl, rows = C.c_void_p(), C.c_int64(), C.c_int32()
        self._ck(self._lib.uda_head_outputs_device(self._h, int(level), int(which), C.byref(ptr), C.byref(fl), C.byref(rows)),
                 "uda_head_outputs_device")
        return ptr.value, fl.value, rows.value

    # ------------------------------------------------------------------ debug / parity accessors
    def preprocessed(self):
        n = self._n_last()
        H, W = self.image_size
        imgs = np.empty((n, H, W, 3), np.float32)
        scales = np.empty((n,), np.float32)
        self._ck(self._lib.uda_get_preprocessed(self._h, _ptr(imgs), _ptr(scales)), "uda_get_preprocessed")
        return imgs, scales

    def preprocessed_scales(self, n):
        """(None, image scales [n]) of the last uint8 batch (host-side copy kept by the handle)."""
        scales = np.empty((max(n, self._cap),), np.float32)
        self._ck(self._lib.uda_get_preprocessed(self._h, None, _ptr(scales)), "uda_get_preprocessed")
        return None, scales[:n]

    def _n_last(self):
        return getattr(self, "_last_n", self._cap)

    def read_buffer(self, name, n_chunk):
        """Activation `name` (plan.buffer_names) of the last chunk: [rows, H, W, C]."""
        bi = self.plan.buffer_names[name]
        b = self.plan.bufs[bi]
        rows = n_chunk * (self.T if b.per_sample else 1)
        out = np.empty((rows, b.H, b.W, b.C), np.float32)
        self._ck(self._lib.uda_read_buffer(self._h, bi, _ptr(out), out.size), "uda_read_buffer")
        return out

    def candidates(self, n):
        K = self._lib.uda_num_candidates(self._h)
        boxes = np.empty((n, K, 4), np.float32)
        scores = np.empty((n, K), np.float32)
        classes = np.empty((n, K), np.int32)
        ucls = np.zeros((n, K, 1 if self.params["nms_configs"].get("max_nms_inputs", 0) else self.num_classes), np.float32)
        ual = np.zeros((n, K, 4), np.float32)
        uep = np.zeros((n, K, 4), np.float32)
        self._ck(self._lib.uda_get_candidates(self._h, _ptr(boxes), _ptr(scores), _ptr(classes), _ptr(ucls),
                                              _ptr(ual), _ptr(uep)), "uda_get_candidates")
        return dict(boxes=boxes, scores=scores, classes=classes, u_cls=ucls, u_al=ual, u_ep=uep)

    def dropout_masks(self, n):
        """{site: [N, T, C]} masks used by the last run."""
        total = sum(ch for _, ch, _ in self.plan.sites) * n * self.T
        flat = np.empty(total, np.float32)
        self._ck(self._lib.uda_get_dropout_masks(self._h, _ptr(flat), flat.size), "uda_get_dropout_masks")
        out, off = {}, 0
        for name, ch, _ in self.plan.sites:
            sz = n * self.T * ch
            out[name] = flat[off:off + sz].reshape(n, self.T, ch)
            off += sz
        return out

    def nms(self, boxes, scores, max_out=100, iou_thresh=0.5, score_thresh=0.001, soft_sigma=0.25, pad=True):
        """NonMaxSuppressionV5 kernel on host arrays: boxes [n,k,4], scores [n,k]."""
        boxes = np.ascontiguousarray(boxes, dtype=np.float32)
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        n, k = scores.shape
        idx = np.zeros((n, max_out), np.int32)
        sc = np.zeros((n, max_out), np.float32)
        valid = np.zeros((n,), np.int32)
        self._ck(self._lib.uda_nms(self._h, _ptr(boxes), _ptr(scores), n, k, max_out, iou_thresh, score_thresh,
                                   soft_sigma, int(pad), _ptr(idx), _ptr(sc), _ptr(valid)), "uda_nms")
        return idx, sc, valid

    def nms_prefix_fallbacks(self):
        """Images / NMS problems so far whose score prefix failed the device check and were redone on the full
        candidate set (identical results either way; include/uda_hip.h uda_nms_prefix_fallbacks)."""
        return int(self._lib.uda_nms_prefix_fallbacks(self._h))

    def nms_coop_not_launched(self):
        """NMS runs that wanted the single-launch grid and did not get it (include/uda_hip.h uda_nms_coop_not_launched):
        0 in normal operation."""
        return int(self._lib.uda_nms_coop_not_launched(self._h))

    def nms_coop_fallbacks(self):
        """Post-process runs redone with two launches per epoch because the single-launch NMS grid timed out
        (include/uda_hip.h uda_nms_coop_fallbacks); 0 in normal operation."""
        return int(self._lib.uda_nms_coop_fallbacks(self._h))

    # ------------------------------------------------------------------ resident-input fast path (bench)
    def stage_images(self, image_arrays):
        """Upload uint8 images once (the PCIe leg); `run_resident` then re-runs the path on them."""
        n = self._feed(image_arrays)
        self._ck(self._lib.uda_synchronize(self._h), "uda_synchronize")
        self._last_n = n
        return n

    def prefetch_images(self, image_arrays):
        """Start the upload of the NEXT batch into the second input slot (copy stream); returns at once."""
        self._prefetched_n = self._feed(image_arrays, prefetch=True)
        return self._prefetched_n

    def swap_prefetched(self):
        """The prefetched batch becomes the input of the next run (the compute stream waits for its upload event)."""
        self._ck(self._lib.uda_swap_prefetched(self._h), "uda_swap_prefetched")
        self._last_n = self._prefetched_n
        return self._last_n

    def run_resident(self, sync=True):
        self._next_seed()
        self._run_id += 1
        self._ck(self._lib.uda_run(self._h, -1, 1), "uda_run")
        if sync:
            self._ck(self._lib.uda_synchronize(self._h), "uda_synchronize")

    def synchronize(self):
        self._ck(self._lib.uda_synchronize(self._h), "uda_synchronize")

    def profile_enable(self, kinds):
        mask = 0
        for k in kinds:
            mask |= 1 << k
        self._ck(self._lib.uda_profile_enable(self._h, mask), "uda_profile_enable")

    def profile_read(self, kind, reset=True):
        ms, cnt = C.c_double(), C.c_int64()
        self._ck(self._lib.uda_profile_read(self._h, kind, C.byref(ms), C.byref(cnt), int(reset)), "uda_profile_read")
        return ms.value, cnt.value

    # ------------------------------------------------------------------ benchmark (infer_lib.py:206-230)
    def _benchmark(self, image_arrays, test_func, bm_runs=10, trace_filename=None):
        """3 warm-up calls, `bm_runs` timed calls; prints per-batch latency and FPS (infer_lib.py:206-230)."""
        for _ in range(3):
            test_func(image_arrays)
        start = time.perf_counter()
        for _ in range(bm_runs):
            test_func(image_arrays)
        end = time.perf_counter()
        inference_time = (end - start) / bm_runs
        print("Per batch inference time: ", inference_time)
        print("FPS: ", (self.batch_size or 1) / inference_time)
        if trace_filename:
            # the reference writes a TF profiler trace of one more call; here: per-op-kind device time of one more call
            kinds = list(range(1, 9)) + [capi.PROF_AGGREGATE, capi.PROF_NMS, capi.PROF_PREPROCESS]
            self.profile_enable(kinds)
            test_func(image_arrays)
            self.synchronize()
            import json
            with open(trace_filename, "w") as f:
                json.dump({str(k): dict(zip(("ms", "launches"), self.profile_read(k))) for k in kinds}, f)
            self.profile_enable([])
        return inference_time

    def benchmark(self, image_arrays, bm_runs=10, trace_filename=None):
        return self._benchmark(image_arrays, self.predict, bm_runs, trace_filename)

    def visualize(self, image, boxes, classes, scores, uncertainty=None, **kwargs):
        """Visualize prediction on image (infer_lib.py:194-204): host drawing, not on the hot path."""
        from .visualize import visualize_image
        return visualize_image(image, boxes, np.asarray(classes).astype(int), scores, self.label_map, uncertainty, **kwargs)


def read_images(paths):
    """Decode image files into uint8 RGB arrays [h, w, 3] (host side, PIL) - the `np.array(Image.open(f))` of the
    reference's callers (validate_model.py:479-483, infer_model.py:554-560).  Sizes may differ; `serve` takes the list."""
    from PIL import Image
    out = []
    for f in paths:
        with Image.open(f) as im:
            out.append(np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8)))
    return out


class KerasDriver(ServingDriver):
    """infer_lib.py:416-491: `KerasDriver(ckpt_path, debug, model_name, batch_size, only_network, model_params)`."""

    def __init__(self, ckpt_path, debug, *args, **kwargs):
        kwargs.setdefault("weights_path", ckpt_path)
        super().__init__(*args, **kwargs)
        self.debug = debug

    def export(self, *args, **kwargs):
        raise NotImplementedError("SavedModel / TFLite / TensorRT export is TensorFlow tooling outside the hot path "
                                  "(infer_lib.py:493-616); save the weight set with weights.save_weights instead")


class SavedModelDriver(ServingDriver):
    """infer_lib.py:299-350: `SavedModelDriver(saved_model_dir_or_frozen_graph, model_name, batch_size, only_network,
    model_params)`.  The path names a weight set here (module docstring): there is no TF graph to load."""

    def __init__(self, saved_model_dir_or_frozen_graph, *args, **kwargs):
        kwargs.setdefault("weights_path", saved_model_dir_or_frozen_graph)
        super().__init__(*args, **kwargs)


class TfliteDriver(ServingDriver):
    """infer_lib.py:353-413.  TFLite flatbuffers are not served by the HIP path."""

    def __init__(self, tflite_path, *args, **kwargs):
        raise ValueError("a .tflite model (%s) cannot be served by the HIP path: pass a checkpoint / .npz weight set" % (tflite_path,))


class DeviceHeads:
    """The five per-level head outputs of a driver's last run, left on the device.

    Behaves as the list the reference's model returns (len 5, indexable, iterable; elements [N,h,w,ch] or
    [T,N,h,w,ch]); the download happens on first access.  `ServingDriver.postprocess` and
    `postprocess.generate_detections` recognise it and post-process in place, so the eval flow
    `cls, box = mc_eval(model, images, config); generate_detections(config, cls, box, ...)` (eval.py:108-123)
    never moves the 700 MB of head outputs through the host."""

    def __init__(self, driver, n, which):
        self.driver, self.n, self.which, self.run_id = driver, int(n), int(which), driver._run_id
        self._host = None

    def _fetch(self):
        if self._host is None:
            if self.run_id != self.driver._run_id:
                raise RuntimeError("these head outputs were overwritten by a later run of the driver")
            self._host = self.driver.head_outputs(self.n)[self.which]
        return self._host

    def __len__(self):
        return len(self.driver.plan.level_hw)

    def __getitem__(self, i):
        return self._fetch()[i]

    def __iter__(self):
        return iter(self._fetch())


class EnsembleDriver:
    """Deep ensemble of M independently initialised / trained detectors (BASELINE configs[3]).

    The reference has no ensemble code: this is the build's extension, defined as
    "aggregate the members exactly like MC samples" — class logits: mean / population std over
    members (a8); boxes: per-member decode, mean / std of the corners, mean of the decoded sigma
    (a14); then the same NMS.  Each member is a deterministic network (no dropout) on its own
    handle; their head outputs are copied device-to-device into the sample slots of a
    post-processing handle whose sample axis has length M.
    """

    def __init__(self, member_weights, model_name="efficientdet-d0", batch_size=1, model_params=None, device=0,
                 chunk_images=None):
        params = dict(model_params or {})
        if params.get("mc_dropout"):
            raise ValueError("ensemble members are deterministic networks: mc_dropout must be off")
        self.members = [ServingDriver(model_name, batch_size=batch_size, model_params=params, weights=w,
                                      device=device, chunk_images=chunk_images) for w in member_weights]
        M = len(self.members)
        if M < 2:
            raise ValueError("an ensemble needs at least two members")
        # the aggregator only post-processes: it is planned as an M-sample MC model of the same geometry
        post_params = dict(params, mc_dropout=True, mc_dropoutrate=1e-9, mc_dropoutsamp=M)
        self.post = ServingDriver(model_name, batch_size=batch_size, model_params=post_params,
                                  weights=member_weights[0], device=device, chunk_images=1)
        self.params = self.post.params
        self.batch_size = batch_size

    def serve(self, image_arrays, post_mode=None):
        lib = self.post._lib
        shared = None           # (device pointer, n, h, w) of member 0's uploaded batch: ONE upload, the others copy device-to-device
        for m, drv in enumerate(self.members):
            if m == 0 or shared is None:
                n = drv._feed(image_arrays)
                if m == 0 and all(d.device == drv.device for d in self.members):
                    ptr, nn, hh, ww = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
                    if lib.uda_input_u8_device(drv._h, C.byref(ptr), C.byref(nn), C.byref(hh), C.byref(ww)) == 0:   # (one raw size)
                        shared = (ptr, nn.value, hh.value, ww.value)
            else:
                drv._ck(lib.uda_set_images_u8_device(drv._h, *shared), "uda_set_images_u8_device")
            drv._ck(lib.uda_run(drv._h, -1, 0), "uda_run")
            self.post._ck(lib.uda_copy_heads(self.post._h, drv._h, n, m), "uda_copy_heads")
        self.post._run_id += 1
        _, scales = self.members[0].preprocessed_scales(n)
        mode = self.post._mode(post_mode)
        s = np.ascontiguousarray(scales, dtype=np.float32)
        self.post._ck(lib.uda_postprocess_heads(self.post._h, n, _ptr(s), mode), "uda_postprocess_heads")
        return self.post._collect(n, mode)

    def close(self):
        for d in self.members + [self.post]:
            d.close()
