"""Multi-GPU exchange steps of the render path, one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).

Two partitions of the work (the scene is replicated on every GPU):

  tile partition    rank r renders the 8-row-aligned stripes s with s % world == r of every frame
                    (ptr_set_partition); pixels never interact, so the merged image is bit-identical to the
                    single-GPU frame.  Exchange: ONE all-gather of the owned stripes (frame / world bytes per rank).
  sample partition  rank r renders whole frames for the global samples r+1, r+1+world, ... (ptr_set_sample_partition)
                    and keeps the running mean of its own samples.  Exchange: all-reduce(sum) of the linearized
                    means, then resolve.  Not bit-equal to sequential accumulation (the per-frame clamp and sRGB
                    round trip of the accumulate step are nonlinear): checked by tolerance only.
"""
import torch
import torch.distributed as dist


def owned_stripes(rank, world, stripe_rows, height):
    """Indices of the stripes (of `stripe_rows` image rows) that `rank` renders."""
    n_stripes = (height + stripe_rows - 1) // stripe_rows
    return list(range(rank, n_stripes, world))


def owned_row_mask(rank, world, stripe_rows, height):
    rows = torch.arange(height)
    return ((rows // stripe_rows) % world) == rank


def _padded_view(image, stripe_rows, world):
    """image: (H, W, 4).  Returns (padded (S, stripe_rows, W, 4) tensor, S) with S a multiple of world."""
    H, W, C = image.shape
    n_stripes = (H + stripe_rows - 1) // stripe_rows
    n_padded = ((n_stripes + world - 1) // world) * world
    pad_rows = n_padded * stripe_rows - H
    if pad_rows:
        image = torch.cat([image, image.new_zeros((pad_rows, W, C))], dim=0)
    return image.view(n_padded, stripe_rows, W, C), n_padded


def gather_tiles(image, rank, world, stripe_rows, group=None):
    """All-gathers the stripes each rank owns into the full frame.  `image` is this rank's (H, W, 4) frame whose
    owned rows are valid; returns the merged (H, W, 4) frame (a new tensor), identical on every rank."""
    if world == 1 and not dist.is_initialized():
        return image.clone()
    H = image.shape[0]
    stripes, n_padded = _padded_view(image, stripe_rows, world)
    mine = stripes[rank::world].contiguous()                      # (n_padded / world, stripe_rows, W, 4)
    if image.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path (gloo has no device collectives here): stage through host memory
        parts = [torch.empty(mine.shape, dtype=image.dtype) for _ in range(world)]
        dist.all_gather(parts, mine.cpu(), group=group)
        gathered = torch.stack(parts).to(image.device)
    else:
        gathered = torch.empty((world,) + tuple(mine.shape), dtype=image.dtype, device=image.device)
        dist.all_gather_into_tensor(gathered.view(-1), mine.view(-1), group=group)
    # gathered[r, j] is stripe j * world + r
    merged = gathered.permute(1, 0, 2, 3, 4).reshape(n_padded * stripe_rows, image.shape[1], image.shape[2])
    return merged[:H].contiguous()


class _DeviceMemory:
    """A library buffer seen through the CUDA array interface, so that torch can wrap it without a copy."""

    def __init__(self, address, floats):
        self.__cuda_array_interface__ = dict(shape=(floats,), typestr="<f4", data=(address, False), version=2)


class TileExchange:
    """The tile partition's exchange on the library's own kernels: ptr_image_pack_stripes -> ONE all-gather ->
    ptr_image_unpack_stripes, in two device buffers made once (three launches and no temporary per exchange, where
    gather_tiles builds the chunks with torch views and copies).  Everything runs on torch's current stream, which must be
    the context's stream."""

    def __init__(self, ctx, width, height, rank, world, stripe_rows):
        self.ctx, self.rank, self.world, self.stripe_rows = ctx, rank, world, stripe_rows
        stripes = (height + stripe_rows - 1) // stripe_rows
        self.chunk_floats = ((stripes + world - 1) // world) * stripe_rows * width * 4
        self.packed = ctx.create_buffer(self.chunk_floats * 4)
        self.gathered = ctx.create_buffer(self.chunk_floats * 4 * world)
        self.packed_t = torch.as_tensor(_DeviceMemory(self.packed.device_address, self.chunk_floats), device="cuda")
        self.gathered_t = torch.as_tensor(_DeviceMemory(self.gathered.device_address, self.chunk_floats * world), device="cuda")

    def merge(self, texture, group=None):
        """All-gathers the stripes each rank owns of `texture` (an RGBA32F Texture of the context) into that same image."""
        self.ctx.pack_stripes(texture, self.rank, self.world, self.stripe_rows, self.packed)
        if dist.get_backend(group) == "gloo":                      # rehearsal path: stage through host memory
            parts = [torch.empty(self.chunk_floats) for _ in range(self.world)]
            dist.all_gather(parts, self.packed_t.cpu(), group=group)
            self.gathered_t.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(self.gathered_t, self.packed_t, group=group)
        self.ctx.unpack_stripes(texture, self.world, self.stripe_rows, self.gathered)


def allreduce_linear(image_linear, group=None):
    """Sums the linearized per-GPU means in place (alpha carries the number of contributions)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if image_linear.is_cuda and dist.get_backend(group) == "gloo":
            # rehearsal path (gloo has no device collectives here): stage through host memory
            staged = image_linear.cpu()
            dist.all_reduce(staged, op=dist.ReduceOp.SUM, group=group)
            image_linear.copy_(staged)
        else:
            dist.all_reduce(image_linear, op=dist.ReduceOp.SUM, group=group)
    return image_linear
