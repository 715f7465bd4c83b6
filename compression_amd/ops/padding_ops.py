"""`same_padding_for_kernel` (python/ops/padding_ops.py:22-51)."""

__all__ = ["same_padding_for_kernel"]


def same_padding_for_kernel(shape, corr, strides_up=None):
    """Padding (before, after) per dimension so that output[i] aligns with input[i*stride].

    corr=True: the correlation kernel is centred at shape // 2;  corr=False (true
    convolution): mirrored.  With strides_up, the padding is that of the equivalent
    zero-upsampled convolution."""
    rank = len(shape)
    if strides_up is None:
        strides_up = rank * (1,)
    if corr:
        padding = [(s // 2, (s - 1) // 2) for s in shape]
    else:
        padding = [((s - 1) // 2, s // 2) for s in shape]
    padding = [((padding[i][0] - 1) // strides_up[i] + 1,
                (padding[i][1] - 1) // strides_up[i] + 1) for i in range(rank)]
    return padding
