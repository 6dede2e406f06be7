"""Config schema: every key the reference declares for the matching path (config/default.py:3-115;
regression / training keys are kept so that the reference's dataset yamls still merge), plus the
keys this implementation adds (declared here because unknown keys are rejected on merge):

  FEATURE_MATCHING   additionally accepts 'SuperGlue' (online SuperPoint+SuperGlue on the GPU)
  DATASET.SYNTHETIC  explicit opt-in to the synthetic stand-in dataset (a missing DATA_ROOT is otherwise an error)
  RANSAC.SEED        seed of the counter-based RANSAC RNG (the reference has no seed knob)
  HIP.*              batch size / keypoint budget of the fused device pipeline; GRAPH_BATCH1: replay the batch-1 online
                     matcher of the per-pair plugin API from one captured HIP graph (nets/graph.py); GRAPH_FUSED: the same for the
                     matcher stage of the fused batched pipeline (opt-in: batches of >= 8 pairs are GPU-bound, measured no gain);
                     EMAT_SCORE: model quality of the E-matrix RANSAC -- 'magsac' (MAGSAC++ loss + sigma-consensus++, the method the
                     reference asks OpenCV for: cv.USAC_MAGSAC, pose_solver.py:46-48) | 'count' (inlier count + LM polish, rounds 1-3);
                     MAGSAC_MAX_THR_RATIO: k * sigma_max of MAGSAC++ as a multiple of EMAT_RANSAC.PIX_THRESHOLD (>= 1);
                     REF_FEATURE_CACHE: the fused pipeline runs SuperPoint once per distinct reference view of a batch (and keeps the
                     last few across batches) instead of once per pair -- same bits, fewer images (pipeline.py);
                     LOADER_WORKERS: decode workers per rank, 0 = the CPUs the container grants divided by the node's ranks (datasets.usable_cpus);
                     LOADER_DECODE: 'process' | 'thread' -- who decodes the JPEG / PNG files of predict_fused's batches: forked worker
                     processes writing into pinned shared-memory batch slots (default; 820 vs 437 pairs/s on the 16-CPU GPU box,
                     profiles/r04_fused_split_1gpu_sg_pnp_{process,thread}.json) or a thread pool under one interpreter lock (datasets.py);
                     CONV / CONV_KERNEL / FUSED_CONV_RELU / RPR_CONV / RPR_CONV_BWD / RPR_CONV_ORDER / RPR_WGRAD_SPLITS: which of two
                     implementations of a layer runs (A/B measurement, parity tests) -- options.py lists values and defaults
  LOFTR.WEIGHTS      checkpoint of the online LoFTR matcher ('LoFTR' feature matching)
  ALLOW_SYNTHETIC_WEIGHTS  hand out seeded synthetic network weights when no checkpoint is configured (tests / benches)
  TRAINING.PRECISION 'bf16' (autocast; the aggregator kernel and the pose algebra stay fp32) | 'fp32'
  TRAINING.SIAMESE_BATCH  encode both images of a pair in one encoder pass; BatchNorm keeps per-view statistics (= the two-call arithmetic)
  TRAINING.DDP_BUCKET_MB  gradient all-reduce bucket size
  TRAINING.GRAPH_STEP     forward + loss + backward of the training step replayed from one captured HIP graph; gradients in one flat
                          buffer, one all-reduce per step instead of DDP's bucket hooks (regression/train.py)
  TRAINING.CHANNELS_LAST  keep weights / activations of the regression model NHWC (MIOpen's implicit-GEMM kernels are NHWC)
  SUPERGLUE.*        matcher hyper-parameters of record (matchers.py:65-71) and weight paths
"""
from .node import CfgNode as CN


def get_cfg_defaults():
    c = CN()
    c.MODEL = None
    c.DEBUG = False
    c.ENCODER = CN(); c.ENCODER.TYPE = None; c.ENCODER.NUM_BLOCKS = None; c.ENCODER.BLOCK_TYPE = None
    c.ENCODER.NOT_CONCAT = None; c.ENCODER.NUM_OUT_LAYERS = None
    c.AGGREGATOR = CN()
    for k, v in dict(TYPE=None, POSITION_ENCODER=None, POSITION_ENCODER_IM1=None, MAX_SCORE_CHANNEL=None,
                     NORMALISE_DOT=False, RESIDUAL_ATT=False, CV_OUTLAYERS=0, CV_HALF_CHANNELS=False,
                     UPSAMPLE_POS_ENC=0, DUSTBIN=False).items():
        c.AGGREGATOR[k] = v
    c.HEAD = CN()
    for k, v in dict(TYPE=None, ADD_BASIS=False, NUM_PTS=6, AVG_POOL=False, BATCH_NORM=True, SEPARATE_SCALE=True).items():
        c.HEAD[k] = v
    c.BACKPROJECT_ANCHORS = None
    # feature matching options (config/default.py:39-66)
    c.FEATURE_MATCHING = None      # 'SIFT' | 'Precomputed' | 'SuperGlue' | 'LoFTR' (new: online, on the GPU)
    c.POSE_SOLVER = None           # 'EssentialMatrix' | 'EssentialMatrixMetric' | 'Procrustes' | 'PNP'
    c.SIFT = CN(); c.SIFT.NUM_FEATURES = None; c.SIFT.RATIO_THRESHOLD = None
    c.MATCHES_FILE_PATH = None
    c.EMAT_RANSAC = CN(); c.EMAT_RANSAC.PIX_THRESHOLD = None; c.EMAT_RANSAC.SCALE_THRESHOLD = None
    c.EMAT_RANSAC.CONFIDENCE = None
    c.PROCRUSTES = CN(); c.PROCRUSTES.MAX_CORR_DIST = None; c.PROCRUSTES.REFINE = False
    c.PNP = CN(); c.PNP.RANSAC_ITER = None; c.PNP.REPROJECTION_INLIER_THRESHOLD = None; c.PNP.CONFIDENCE = None
    # dataset (config/default.py:68-92)
    c.DATASET = CN()
    for k, v in dict(DATA_SOURCE=None, SCENES=None, DATA_ROOT=None, NPZ_ROOT=None, MIN_OVERLAP_SCORE=None,
                     MAX_OVERLAP_SCORE=None, AUGMENTATION_TYPE=None, BLACK_WHITE=False, HEIGHT=None, WIDTH=None,
                     ESTIMATED_DEPTH=None, QUERY_FRAME_COUNT=1).items():
        c.DATASET[k] = v
    c.DATASET.SYNTHETIC = None      # [n_scenes, frames_per_scene]: run on the synthetic stand-in ON PURPOSE (no data offline)
    c.DATASET.PAIRS_TXT = CN(); c.DATASET.PAIRS_TXT.TRAIN = None; c.DATASET.PAIRS_TXT.VAL = None
    c.DATASET.PAIRS_TXT.TEST = None; c.DATASET.PAIRS_TXT.ONE_NN = False
    # training (config/default.py:94-112); PRECISION / SIAMESE_BATCH / DDP_BUCKET_MB are this implementation's (regression/train.py)
    c.TRAINING = CN()
    for k, v in dict(BATCH_SIZE=None, NUM_WORKERS=None, SAMPLER=None, N_SAMPLES_SCENE=None,
                     SAMPLE_WITH_REPLACEMENT=None, LR=None, LR_STEP_INTERVAL=None, LR_STEP_GAMMA=None,
                     VAL_INTERVAL=None, VAL_BATCHES=None, LOG_INTERVAL=None, EPOCHS=None, GRAD_CLIP=0.,
                     ROT_LOSS='rot_frobenius_loss', TRANS_LOSS='trans_l2_loss', LAMBDA=1.0,
                     PRECISION='bf16', SIAMESE_BATCH=False, DDP_BUCKET_MB=64, CHANNELS_LAST=False, GRAPH_STEP=False).items():
        c.TRAINING[k] = v
    # ---- additions of this implementation ----
    c.RANSAC = CN(); c.RANSAC.SEED = 0
    c.HIP = CN(); c.HIP.BATCH_PAIRS = 16; c.HIP.MAX_KEYPOINTS = 1024; c.HIP.MAX_CORRESPONDENCES = 8192; c.HIP.GRAPH_BATCH1 = True; c.HIP.GRAPH_FUSED = False
    c.HIP.EMAT_SCORE = 'magsac'; c.HIP.MAGSAC_MAX_THR_RATIO = 1.0; c.HIP.REF_FEATURE_CACHE = True; c.HIP.LOADER_DECODE = 'process'; c.HIP.LOADER_WORKERS = 0
    from .. import options as _opt                     # kernel-selection options (options.py): declared with their defaults, applied by apply_cfg
    for _k in _opt.names():
        c.HIP[_k] = _opt.default(_k)
    c.SUPERGLUE = CN()
    for k, v in dict(NMS_RADIUS=4, KEYPOINT_THRESHOLD=0.005, MAX_KEYPOINTS=1024, SINKHORN_ITERATIONS=20,
                     MATCH_THRESHOLD=0.2, SUPERPOINT_WEIGHTS=None, SUPERGLUE_WEIGHTS=None, SYNTHETIC_SEED=1234).items():
        c.SUPERGLUE[k] = v
    c.LOFTR = CN(); c.LOFTR.WEIGHTS = None
    c.ALLOW_SYNTHETIC_WEIGHTS = False
    return c


cfg = get_cfg_defaults()
