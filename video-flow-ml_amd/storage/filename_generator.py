"""Cache-directory and output-file naming; same strings as reference storage/filename_generator.py
(generate_output_filename :13-99, generate_output_filepath :102-148, generate_cache_directory :151-212).
The cache directory name is how a later run (or the reference's GUI tools) finds the flow fields
this engine wrote, so it is part of the drop-in contract."""
import os
from pathlib import Path
from typing import Optional


def generate_output_filename(input_path: str, start_time: Optional[float] = None,
                             duration: Optional[float] = None, start_frame: int = 0, max_frames: int = 1000,
                             flow_only: bool = False, taa: bool = False, fast_mode: bool = False,
                             tile_mode: bool = False, uncompressed: bool = False, flow_format: str = 'gamedev',
                             motion_vectors_clamp_range: float = 32.0, fps: float = 30.0) -> str:
    parts = [os.path.splitext(os.path.basename(input_path))[0]]
    if start_time is not None:
        parts.append(f"{start_time}s")
    elif start_frame > 0:
        parts.append(f"f{start_frame}")
    if duration is not None:
        parts.append(f"{duration}s")
    elif max_frames != 1000:
        parts.append(f"{max_frames}f")
    parts += [tag for tag, on in (("fast", fast_mode), ("tile", tile_mode)) if on]
    if flow_only:
        if flow_format == 'gamedev':
            parts.append("gamedev")
        else:
            clean = flow_format.replace('-', '_').replace('_flow', '').replace('flow_', '')
            if flow_format.startswith('motion-vectors'):
                clean = f"{clean}_{int(motion_vectors_clamp_range)}"
            parts.append(clean)
    elif taa:
        parts.append("taa")
    parts.append(f"{fps:.0f}fps")
    parts.append("uncompressed_I420" if uncompressed else "MJPG")
    return "_".join(parts) + ".avi"


def generate_output_filepath(input_path: str, output_dir: str, **kwargs) -> str:
    name = generate_output_filename(input_path=input_path, **kwargs)
    if not os.path.exists(output_dir):
        os.makedirs(output_dir)
    return os.path.join(output_dir, name)


def generate_cache_directory(input_path: str, start_frame: int = 0, max_frames: int = 1000,
                             sequence_length: int = 5, fast_mode: bool = False, tile_mode: bool = False,
                             model: str = 'videoflow', dataset: str = 'things', architecture: str = 'mof',
                             variant: str = 'noise') -> str:
    """`{stem}_flow_cache_{model}[_{arch}_{dataset}_{variant} | _{dataset}]_seq{T}_start{s}_frames{n}[_fast][_tile]`
    next to the input video."""
    src = Path(input_path)
    ident = [model]
    if model == 'videoflow':
        ident += [architecture, dataset, variant]
    elif model == 'memflow':
        ident.append(dataset)
    run = [f"seq{sequence_length}", f"start{start_frame}", f"frames{max_frames}"]
    if fast_mode:
        run.append("fast")
    if tile_mode:
        run.append("tile")
    return str(src.parent / f"{src.stem}_flow_cache_{'_'.join(ident)}_{'_'.join(run)}")
