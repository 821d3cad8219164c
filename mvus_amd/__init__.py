"""mvus_amd -- MI355X-native bundle-adjustment core for MultiViewUnsynch-style scenes.

Keeps the reference's ``reconstruction.common.{Scene, Camera, create_scene}`` surface
(``mvus_amd.reconstruction.common``) and runs ``Scene.BA`` / ``Scene.remove_outliers`` on
hand-written HIP kernels (gfx950) through the C ABI declared in ``include/mvus_ba.h``.
"""
__version__ = "0.1.0"
