"""Drop-in replacements for the reference's top-level modules.

Put this directory on sys.path *instead of* the reference's `src/` hot-path modules:

    sys.path.insert(0, os.path.join(os.path.dirname(uav_airvision_amd.__file__), 'dropin'))
    from image_processing import ImageProcessor      # reference: src/image_processing/__init__.py:14-27
    from msckf import MSCKF                          # reference: src/msckf.py:95

so that the reference's `modules/vio.py` (vio.py:3-4,14-15) runs unchanged.
"""
