"""CenterHead configuration used by the g4 fixture and its tests (content of the RADAR_DENSE_HEAD block of the
reference's radar_distill_train.yaml:148-199)."""
HEAD_CFG = dict(
    DISTILL_PRED=True, CLASS_AGNOSTIC=False, IOU_REG="DIoU",
    CLASS_NAMES_EACH_HEAD=[["car"], ["truck", "construction_vehicle"], ["bus", "trailer"], ["barrier"],
                           ["motorcycle", "bicycle"], ["pedestrian", "traffic_cone"]],
    SHARED_CONV_CHANNEL=64, USE_BIAS_BEFORE_NORM=True, NUM_HM_CONV=2,
    SEPARATE_HEAD_CFG=dict(HEAD_ORDER=["center", "center_z", "dim", "rot", "vel", "iou"],
                           HEAD_DICT={k: dict(out_channels=v, num_conv=2) for k, v in
                                      dict(center=2, center_z=1, dim=3, rot=2, vel=2, iou=1).items()}),
    RECTIFIER=0.5,
    TARGET_ASSIGNER_CONFIG=dict(FEATURE_MAP_STRIDE=8, NUM_MAX_OBJS=500, GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2),
    LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=0.25,
                                       code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 1.0, 1.0])),
    # radar_distill_train.yaml:187-195 with K and the score threshold scaled to the 16x16 test map
    POST_PROCESSING=dict(SCORE_THRESH=0.05, POST_CENTER_LIMIT_RANGE=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], MAX_OBJ_PER_SAMPLE=100,
                         NMS_CONFIG=dict(NMS_TYPE="nms_gpu", NMS_THRESH=0.2, NMS_PRE_MAXSIZE=1000, NMS_POST_MAXSIZE=83)),
)
CLASS_NAMES = ["car", "truck", "construction_vehicle", "bus", "trailer", "barrier", "motorcycle", "bicycle",
               "pedestrian", "traffic_cone"]
