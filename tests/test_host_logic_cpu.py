"""Host-side helpers that need no GPU."""
import torch


def test_pack_segments_word_and_byte_paths():
    """train.pack_segments: the pyramid's tensors with their zero pads as one byte buffer; 32-bit concatenation when every
    segment allows it, byte-wise otherwise -- the same bytes either way."""
    from khairil_tum_facade_semantic_segmentation_amd.train import pack_segments
    g = torch.Generator().manual_seed(0)
    tensors = [torch.randn(5, 3, generator=g), None, torch.randint(0, 100, (7,), generator=g), torch.randn(2, 2, generator=g)]
    pads = []
    for t in tensors:
        n = 0 if t is None else t.numel() * t.element_size()
        p = (-n) % 16
        pads.append(torch.zeros(p, dtype=torch.uint8) if p else None)
    flat = pack_segments(tensors, pads)
    assert flat.dtype == torch.uint8 and flat.numel() % 16 == 0
    off = 0
    for t, p in zip(tensors, pads):
        if t is not None:
            n = t.numel() * t.element_size()
            assert torch.equal(flat[off:off + n].view(t.dtype).view(t.shape), t)
            off += n
        if p is not None:
            assert not flat[off:off + p.numel()].any()
            off += p.numel()
    assert off == flat.numel()
    # a segment that is no multiple of four bytes takes the byte path
    odd = [torch.arange(3, dtype=torch.uint8), torch.randn(4, generator=g)]
    flat2 = pack_segments(odd, [None, None])
    assert torch.equal(flat2[:3], odd[0]) and torch.equal(flat2[3:].clone().view(torch.float32), odd[1])
