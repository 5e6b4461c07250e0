"""Package paths and the file-name -> class-name rule used by every registry
(reference: gops/utils/gops_path.py:12-20)."""
import os

gops_path = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
algorithm_path = os.path.join(gops_path, "algorithm")
apprfunc_path = os.path.join(gops_path, "apprfunc")
env_path = os.path.join(gops_path, "env")
trainer_path = os.path.join(gops_path, "trainer")


def underline2camel(s: str, first_upper: bool = False) -> str:
    """`on_serial_trainer` -> `OnSerialTrainer`; with first_upper `fhadp` -> `FHADP`,
    `fhadp_exterior` -> `FHADPExterior`."""
    parts = s.split("_")
    out = parts.pop(0).upper() if first_upper else ""
    return out + "".join(p[:1].upper() + p[1:] for p in parts)
