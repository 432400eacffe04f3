"""Head meta data: the subset of the reference's ``openpifpaf.headmeta`` the decode
path reads (reference ``headmeta.py:12-31,36-104``): keypoints/skeleton, the
field counts and ``stride = base_stride // upsample_stride``."""
from dataclasses import dataclass, field
from typing import Any, ClassVar, List, Optional, Tuple


@dataclass
class Base:
    name: str
    dataset: str

    head_index: Optional[int] = field(default=None, init=False)
    base_stride: Optional[int] = field(default=None, init=False)
    upsample_stride: int = field(default=1, init=False)

    n_confidences: ClassVar[int] = 1
    n_vectors: ClassVar[int] = 1
    n_scales: ClassVar[int] = 1
    vector_offsets: ClassVar[List[bool]] = [True]

    @property
    def stride(self) -> Optional[int]:
        if self.base_stride is None:
            return None
        return self.base_stride // self.upsample_stride

    @property
    def n_fields(self) -> int:
        raise NotImplementedError


@dataclass
class Cif(Base):
    """Composite Intensity Field: one field per keypoint, components
    (width, confidence, x, y, scale)."""
    keypoints: List[str] = None
    sigmas: List[float] = None
    pose: Any = None
    draw_skeleton: Optional[List[Tuple[int, int]]] = None
    score_weights: Optional[List[float]] = None

    n_confidences: ClassVar[int] = 1
    n_vectors: ClassVar[int] = 1
    n_scales: ClassVar[int] = 1
    vector_offsets: ClassVar[List[bool]] = [True]

    decoder_min_scale = 0.0
    decoder_seed_mask: Optional[List[int]] = None

    @property
    def n_fields(self) -> int:
        return len(self.keypoints)


@dataclass
class Caf(Base):
    """Composite Association Field: one field per bone, components
    (width, confidence, x1, y1, x2, y2, scale1, scale2)."""
    keypoints: List[str] = None
    sigmas: List[float] = None
    skeleton: List[Tuple[int, int]] = None      # 1-based joint pairs
    pose: Any = None
    sparse_skeleton: Optional[List[Tuple[int, int]]] = None
    dense_to_sparse_radius: float = 2.0
    only_in_field_of_view: bool = False

    n_confidences: ClassVar[int] = 1
    n_vectors: ClassVar[int] = 2
    n_scales: ClassVar[int] = 2
    vector_offsets: ClassVar[List[bool]] = [True, True]

    decoder_min_distance = 0.0
    decoder_max_distance = float('inf')
    decoder_confidence_scales: Optional[List[float]] = None

    @property
    def n_fields(self) -> int:
        return len(self.skeleton)

    @staticmethod
    def concatenate(metas):
        """Reference ``headmeta.py:89-113``: one meta whose skeleton is the metas' skeletons in a row
        (used by ``CifCafDense``: sparse + dense CAF heads decoded as one CAF head)."""
        concatenated = Caf(
            name='_'.join(m.name for m in metas),
            dataset=metas[0].dataset,
            keypoints=metas[0].keypoints,
            sigmas=metas[0].sigmas,
            pose=metas[0].pose,
            skeleton=[s for meta in metas for s in meta.skeleton],
            sparse_skeleton=metas[0].sparse_skeleton,
            only_in_field_of_view=metas[0].only_in_field_of_view,
        )
        concatenated.decoder_confidence_scales = [
            s for meta in metas
            for s in (meta.decoder_confidence_scales if meta.decoder_confidence_scales
                      else [1.0 for _ in meta.skeleton])]
        concatenated.head_index = metas[0].head_index
        concatenated.base_stride = metas[0].base_stride
        concatenated.upsample_stride = metas[0].upsample_stride
        return concatenated


@dataclass
class TSingleImageCif(Cif):
    """Single-image CIF head of a tracking model (reference ``headmeta.py:136-138``)."""


@dataclass
class TSingleImageCaf(Caf):
    """Single-image CAF head of a tracking model (reference ``headmeta.py:141-143``)."""


@dataclass
class Tcaf(Base):
    """Tracking Composite Association Field: one field per joint, from its position in this frame to its
    position in the previous frame (reference ``headmeta.py:146-186``)."""
    keypoints_single_frame: List[str] = None
    sigmas_single_frame: List[float] = None
    pose_single_frame: Any = None
    draw_skeleton_single_frame: Optional[List[Tuple[int, int]]] = None
    keypoints: Optional[List[str]] = None
    sigmas: Optional[List[float]] = None
    pose: Any = None
    draw_skeleton: Optional[List[Tuple[int, int]]] = None
    only_in_field_of_view: bool = False

    n_confidences: ClassVar[int] = 1
    n_vectors: ClassVar[int] = 2
    n_scales: ClassVar[int] = 2
    vector_offsets: ClassVar[List[bool]] = [True, True]

    def __post_init__(self):
        if self.keypoints is None:
            self.keypoints = list(self.keypoints_single_frame) * 2
        if self.sigmas is None:
            self.sigmas = list(self.sigmas_single_frame) * 2

    @property
    def skeleton(self):
        n = len(self.keypoints_single_frame)
        return [(i + 1, i + 1 + n) for i in range(n)]

    @property
    def n_fields(self) -> int:
        return len(self.keypoints_single_frame)


@dataclass
class CifDet(Base):
    """Composite Intensity Field for detection: one field per category, components
    (width, confidence, x, y, w, h) (reference ``headmeta.py:116-133``)."""
    categories: List[str] = None

    n_confidences: ClassVar[int] = 1
    n_vectors: ClassVar[int] = 2
    n_scales: ClassVar[int] = 0
    vector_offsets: ClassVar[List[bool]] = [True, False]

    decoder_min_scale = 0.0

    @property
    def n_fields(self) -> int:
        return len(self.categories)


def cocokp_metas(upsample_stride=2, base_stride=16):
    """The (Cif, Caf) pair of the reference's ``cocokp`` datamodule
    (reference ``plugins/coco/cocokp.py:68-85``) with the stride the pretrained
    models use (``--cocokp-upsample=2``)."""
    from . import constants
    cif = Cif('cif', 'cocokp', keypoints=constants.COCO_KEYPOINTS, sigmas=constants.COCO_PERSON_SIGMAS,
              pose=constants.COCO_UPRIGHT_POSE, draw_skeleton=constants.COCO_PERSON_SKELETON,
              score_weights=constants.COCO_PERSON_SCORE_WEIGHTS)
    caf = Caf('caf', 'cocokp', keypoints=constants.COCO_KEYPOINTS, sigmas=constants.COCO_PERSON_SIGMAS,
              pose=constants.COCO_UPRIGHT_POSE, skeleton=constants.COCO_PERSON_SKELETON)
    for i, m in enumerate((cif, caf)):
        m.head_index = i
        m.base_stride = base_stride
        m.upsample_stride = upsample_stride
    return cif, caf


def cocokp_dense_metas(upsample_stride=2, base_stride=16):
    """(Cif, Caf, dense Caf): the ``cocokp`` heads plus the CAF head over ``DENSER_COCO_PERSON_CONNECTIONS``
    (reference ``plugins/coco/cocokp.py:68-85`` with ``--cocokp-with-dense``)."""
    from . import constants
    cif, caf = cocokp_metas(upsample_stride, base_stride)
    dcaf = Caf('caf25', 'cocokp', keypoints=constants.COCO_KEYPOINTS, sigmas=constants.COCO_PERSON_SIGMAS,
               pose=constants.COCO_UPRIGHT_POSE, skeleton=constants.DENSER_COCO_PERSON_CONNECTIONS,
               sparse_skeleton=constants.COCO_PERSON_SKELETON, only_in_field_of_view=True)
    dcaf.head_index = 2
    dcaf.base_stride = base_stride
    dcaf.upsample_stride = upsample_stride
    return cif, caf, dcaf


def wholebody_metas(upsample_stride=2, base_stride=16):
    """The (Cif, Caf) pair of the reference's ``wholebody`` datamodule: 133 keypoints, 160 bones
    (reference ``plugins/wholebody/wholebody.py:91-109``, constants ``plugins/wholebody/constants.py``)."""
    from . import constants
    wb = constants.wholebody()
    cif = Cif('cif', 'wholebody', keypoints=wb['keypoints'], sigmas=wb['sigmas'], pose=wb['standing_pose'],
              draw_skeleton=wb['skeleton'], score_weights=wb['score_weights'])
    caf = Caf('caf', 'wholebody', keypoints=wb['keypoints'], sigmas=wb['sigmas'], pose=wb['standing_pose'],
              skeleton=wb['skeleton'])
    for i, m in enumerate((cif, caf)):
        m.head_index = i
        m.base_stride = base_stride
        m.upsample_stride = upsample_stride
    return cif, caf
