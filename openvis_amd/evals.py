"""Result hand-off to the reference's evaluators: the two `instances_to_*_json_video` converters with the
reference's names, inputs and output structure (openvis/data/evals/ytvis_eval.py:258-301, burst_eval.py:177-240).

The reference copies every dense mask to the host, transposes it to Fortran order and calls pycocotools per frame;
here `outputs["pred_masks_rle"]` (MODEL.MASK_FORMER.TEST.OUTPUT_RLE: masks run-length encoded on the GPU,
openvis_amd/csrc/rle.hip) is passed through, and dense `outputs["pred_masks"]` are still accepted (encoded with the same
run-length rules on the host, as the reference does)."""
import numpy as np

from . import rle


def _segmentations(outputs):
    """per instance: list over frames of {"size": [H, W], "counts": str}."""
    if "pred_masks_rle" in outputs:
        return outputs["pred_masks_rle"]
    segs = []
    for m in outputs["pred_masks"]:
        m = np.asarray(m.cpu() if hasattr(m, "cpu") else m)
        segs.append([{"size": [int(f.shape[0]), int(f.shape[1])],
                      "counts": rle.counts_to_string(rle.mask_to_counts(f)).decode("ascii")} for f in m])
    return segs


def rle_area(seg):
    """foreground pixels of one COCO RLE dict (odd-numbered runs)."""
    return int(sum(rle.string_to_counts(seg["counts"])[1::2]))


def instances_to_coco_json_video(inputs, outputs):
    """ytvis_eval.py:258-301: one dict per predicted track (video_id, score, category_id, per-frame RLE segmentations,
    entropy when present)."""
    assert len(inputs) == 1, "More than one inputs are loaded for inference!"
    video_id = inputs[0]["video_id"]
    results = []
    segs = _segmentations(outputs)
    for instance_id, (s, l, m) in enumerate(zip(outputs["pred_scores"], outputs["pred_labels"], segs)):
        res = {"video_id": video_id, "score": s, "category_id": l, "segmentations": m}
        if "pred_entropys" in outputs:
            res["entropy"] = outputs["pred_entropys"][instance_id]
        results.append(res)
    return results


def instances_to_burst_json_video(inputs, outputs):
    """burst_eval.py:177-240: one sequence dict; a track contributes to a frame only where its mask has more than 20
    pixels (:203), keyed by the instance id, and its category to `track_category_ids`."""
    assert len(inputs) == 1, "More than one inputs are loaded for inference!"
    burst_results = {k: inputs[0][k] for k in ["width", "height", "seq_name", "dataset", "annotated_image_paths"]}
    video_length = inputs[0]["length"]
    burst_results["segmentations"] = [{} for _ in range(video_length)]
    burst_results["track_category_ids"] = {}
    segs = _segmentations(outputs)
    for instance_id, (s, l, m, e) in enumerate(zip(outputs["pred_scores"], outputs["pred_labels"], segs,
                                                   outputs["pred_entropys"])):
        for t, seg in enumerate(m):
            if rle_area(seg) > 20:
                burst_results["segmentations"][t][instance_id] = {"rle": seg["counts"], "is_gt": False, "score": s, "entropy": e}
                burst_results["track_category_ids"][instance_id] = l
    return [burst_results]
