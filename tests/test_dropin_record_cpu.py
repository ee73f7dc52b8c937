"""dropin.record()'s configuration translation (CPU): the MODEL block of GLENet_VR.yaml as the reference's loader leaves it ->
the constants of glenet_amd.glenet_vr; anything the recorded step does not implement is refused."""
import copy

import pytest

from test_dropin_record_gpu import MODEL_CFG


def test_translated_configuration_is_the_packages():
    from glenet_amd import dropin
    from glenet_amd import glenet_vr as gvr
    full = copy.deepcopy(MODEL_CFG)
    full["ROI_HEAD"]["NMS_CONFIG"]["TRAIN"].update(NMS_PRE_MAXSIZE=9000, NMS_POST_MAXSIZE=512)
    full["ROI_HEAD"]["TARGET_CONFIG"]["ROI_PER_IMAGE"] = 128
    full["ROI_HEAD"]["DP_RATIO"] = 0.3
    roi_cfg, head_cfg = dropin._translate_cfg(full)
    assert roi_cfg == gvr.ROI_HEAD_CFG and head_cfg == gvr.DENSE_HEAD_CFG
    with pytest.raises(NotImplementedError):
        dropin._translate_cfg(dict(full, NAME="PVRCNN"))
    multi = copy.deepcopy(full)
    multi["ROI_HEAD"]["NMS_CONFIG"]["TRAIN"]["MULTI_CLASSES_NMS"] = True
    with pytest.raises(NotImplementedError):
        dropin._translate_cfg(multi)




def test_translated_post_processing_is_the_packages():
    from glenet_amd import detector as det
    from glenet_amd import dropin
    full = copy.deepcopy(MODEL_CFG)
    full["POST_PROCESSING"].update(SCORE_THRESH=0.3, POST_SCORE_THRESH=0.81)
    cfg, thresh = dropin._translate_post_cfg(full)
    assert cfg == det.POST_PROCESSING_CFG and thresh == [0.3, 0.5, 0.7]
    other = copy.deepcopy(full)
    other["POST_PROCESSING"]["NMS_CONFIG"]["NMS_TYPE"] = "nms_gpu"
    with pytest.raises(NotImplementedError):
        dropin._translate_post_cfg(other)
