"""MI355X-native TSM training hot path for background-debiased video class-incremental learning.

Import as ``bdvcil_amd`` (see ``bdvcil_amd.py`` at the repo root).  Layout:

* ``csrc/``        hand-written HIP kernels for gfx950 + the C ABI (``include/bdvcil_hip.h``)
* ``_lib``         ctypes binding (lazy, per process, no fallback)
* ``kernels``      tensor-level wrappers (shape checks + launch on torch's current stream)
* ``functional``   autograd glue (block-level Functions)
* ``registry``, ``recognizer``, ``resnet_tsm``, ``heads``, ``losses``, ``hooks``, ``optim``
                   the reference's mmaction2-style plugin surface (same names / signatures / state_dict keys)
* ``frontend``     fused background-mix + normalize; train front-end = RandAugment -> mix decision -> blend
* ``augment``      RandAugment for batches of uint8 clips (draws + device tables; pixels in csrc/augment.hip)
* ``cil_step``     training-step arithmetic of BaseCIL / ICARLModel + a step engine
* ``ddp``          bucketed gradient all-reduce over RCCL
* ``representation``  predict_step / NME classifier / class means / herding
* ``resnet3d``     I3D-ResNet50 (ResNet3d / I3DHead / Recognizer3D) on the same kernels (temporal convs as k x 1 convs)
* ``task_loop``    the CIL task loop (CILTrainer + CILDataModule bookkeeping, same files on disk)
"""
import os as _os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue run in order.  The
# step uses the main stream, the weight-gradient side stream, a pinned-staging copy stream and -- under torch.distributed -- the
# reducer's communication stream plus RCCL's own: with four queues the side stream then shares one, and the overlap of the weight
# gradients with the BatchNorm / dgrad chain is lost (one MI355X, the RCCL path as a one-rank group: 541.8 clips/s against 573.1
# without the process group; with eight queues 571.3 against 574.6).  Must be in the environment before the HIP runtime starts:
# import this package (or set it) before the first torch.cuda call.  An explicit setting of the caller's wins.
if 'GPU_MAX_HW_QUEUES' not in _os.environ:
    _os.environ['GPU_MAX_HW_QUEUES'] = '8'
    import sys as _sys
    _torch = _sys.modules.get('torch')
    if _torch is not None and _torch.cuda.is_initialized():
        # too late for this process: the runtime read its environment when it started (a caller that touched torch.cuda first, or a
        # profiler that preloads the runtime).  Not fatal -- only the stream overlap of the distributed path is at stake.
        import warnings as _warnings
        _warnings.warn('bdvcil_amd: the HIP runtime was initialised before this package was imported, so GPU_MAX_HW_QUEUES=8 cannot '
                       'take effect; under torch.distributed the weight-gradient stream then shares a hardware queue (about 5 % of '
                       'step time). Export GPU_MAX_HW_QUEUES=8 or import bdvcil_amd before the first torch.cuda call.',
                       RuntimeWarning, stacklevel=2)
from .kernels import bump_weight_epoch  # noqa: F401,E402

from . import _lib, kernels  # noqa: F401,E402
from .registry import (BACKBONES, HEADS, LOSSES, OPTIMIZER_BUILDERS, RECOGNIZERS, Registry, build_backbone,  # noqa: F401
                       build_head, build_loss, build_model)
from .resnet_tsm import Nhwc4Frames, ResNetTSM, TemporalShift  # noqa: F401
from .heads import LSC, AvgConsensus, IncrementalNet, IncrementalTSMHead  # noqa: F401
from .losses import ACMSmoothCE, CrossEntropyLoss, LSCLoss, SoftTargetCrossEntropy  # noqa: F401
from .recognizer import CILRecognizer2D, Recognizer2D  # noqa: F401
from .resnet3d import I3DHead, Recognizer3D, ResNet3d  # noqa: F401
from .hooks import OutputHook, rgetattr  # noqa: F401
from .optim import (CILTSMOptimizerConstructor, CILTSMOptimizerConstructorImprovised, FusedSGD, build_lr_scheduler,  # noqa: F401
                    build_optimizer)
from .frontend import BackgroundCropFrontEnd, BackgroundMixFrontEnd, CropFrontEnd, MultiScaleCropResize, TrainClipFrontEnd, crop_offsets  # noqa: F401
from .decode import JpegDecoder, PrefetchLoader, RawFrameClipLoader, sample_frames  # noqa: F401
from .augment import RandAugment  # noqa: F401
from .cil_step import (TrainEngine, base_training_step, icarl_training_step, icarl_video_mix_training_step,  # noqa: F401
                       tubemix_draw)
from .ddp import GradAllReducer, broadcast_parameters  # noqa: F401
from .representation import Herding, ReprPredictor, class_means_from_repr, nme_classify  # noqa: F401
from .task_loop import CILTaskLoop, CILWorkDir, RawframeRecords, SyntheticClipLoader, TaskSplits  # noqa: F401

__version__ = '0.1.0'
