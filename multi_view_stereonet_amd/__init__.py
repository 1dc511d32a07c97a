"""MI355X-native plane-sweep forward of MultiViewStereoNet."""
__all__ = ["MultiViewStereoNet"]


def __getattr__(name):
    if name == "MultiViewStereoNet":
        from .multi_view_stereonet import MultiViewStereoNet
        return MultiViewStereoNet
    raise AttributeError(name)
