"""Time units and item arithmetic (pure integer/float host logic).

Semantics follow the reference's ``mimikit/features/item_spec.py``: units
``Sample/Frame/Second/Step`` (:16-55), ``convert`` (:58-112) and ``ItemSpec``
addition / ``to`` (:115-151).  They size the generate loop
(``GenerateLoopV2.get_n_steps``), the STFT length fix-up and the batch items.
"""
import dataclasses as dtc
from typing import Any, Optional, Union

__all__ = ["Sample", "Frame", "Step", "Second", "Unit", "ItemSpec", "convert"]

# coarser units sort after finer ones; `min` of two units is the conversion target
_RANK = {"Sample": 0, "Frame": 1, "Second": 2, "Step": 3}


class _Unit:
    def __lt__(self, other):
        return _RANK[type(self).__name__] < _RANK[type(other).__name__]

    def __hash__(self):
        return hash(repr(self))


@dtc.dataclass(eq=True)
class Sample(_Unit):
    sr: Optional[int]
    __hash__ = _Unit.__hash__


@dtc.dataclass(eq=True)
class Frame(_Unit):
    frame_size: int
    hop_length: int
    padding: Optional[Any] = None
    __hash__ = _Unit.__hash__

    @property
    def overlap_extra(self) -> int:
        """samples a run of frames needs beyond n_frames * hop (0 when padded/centered)"""
        return 0 if self.padding else self.frame_size - self.hop_length


@dtc.dataclass(eq=True)
class Second(_Unit):
    sr: Optional[int]
    __hash__ = _Unit.__hash__


@dtc.dataclass(eq=True)
class Step(_Unit):
    __hash__ = _Unit.__hash__


Unit = Union[Sample, Frame, Second, Step]


def _common_sr(a, b) -> int:
    found = {u.sr for u in (a, b) if getattr(u, "sr", None) is not None}
    assert len(found) == 1, f"couldn't find a single sr: {a}, {b}"
    return found.pop()


def convert(x: Union[int, float], from_unit: Unit, to_unit: Unit, as_length: bool):
    """Convert a position (``as_length=False``) or a duration (``as_length=True``)
    between units.  Only durations account for the ``frame_size - hop`` overlap of
    un-padded frames."""
    src, dst = type(from_unit), type(to_unit)

    if src is Sample:
        if dst is Frame:
            extra = to_unit.overlap_extra if as_length else 0
            return int((x - extra) // to_unit.hop_length)
        if dst is Second:
            return x / _common_sr(from_unit, to_unit)
        return x

    if src is Frame:
        extra = from_unit.overlap_extra if as_length else 0
        n = x - int(bool(from_unit.padding))
        if dst is Sample:
            return int(n * from_unit.hop_length) + extra
        if dst is Second:
            return (n * from_unit.hop_length + extra) / to_unit.sr
        return n

    if src is Second:
        if dst is Frame:
            extra = to_unit.overlap_extra if as_length else 0
            return (int(x * from_unit.sr) - extra) // to_unit.hop_length
        if dst is Sample:
            return int(x * _common_sr(to_unit, from_unit))
        if dst is Step:
            raise TypeError("can not convert seconds to steps")
        return None

    if src is Step:
        if dst is Step:
            raise TypeError("can not convert steps to seconds")
        return x
    return None


@dtc.dataclass
class ItemSpec:
    shift: Union[int, float] = 0
    length: Union[int, float] = 0
    stride: Union[int, float] = 1
    unit: Unit = dtc.field(default_factory=Step)

    def to(self, unit: Unit) -> "ItemSpec":
        return ItemSpec(
            shift=convert(self.shift, self.unit, unit, as_length=False),
            length=convert(self.length, self.unit, unit, as_length=True),
            stride=self.stride,
            unit=unit,
        )

    def __add__(self, other: "ItemSpec") -> "ItemSpec":
        if not isinstance(other, ItemSpec):
            raise TypeError(f"Expected other to be of type ItemSpec. Got {type(other)}")
        if type(self.unit) is type(other.unit) and self.unit != other.unit:
            raise ValueError("Can not add unit of the same type parametrized differently:\n"
                             f" {self.unit} and {other.unit}")
        finest = min(self.unit, other.unit)
        a = self if self.unit == finest else self.to(finest)
        b = other if other.unit == finest else other.to(finest)
        return ItemSpec(a.shift + b.shift, a.length + b.length, max(a.stride, b.stride), finest)
