"""Small helpers with the reference's names (util.py:10-42).  The reference's two dead functions
(max_batch_size_for_sample_rate, convert_modules: util.py:44-60 use undefined names) are not carried."""
import torch


def add_slash(path):
    if path is None:
        return None
    return path if path.endswith("/") else path + "/"


def denorm_celeba(img):
    return ((img + 1) / 2).clamp(0, 1)


def make_grid(images, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid (torchvision is absent; its published layout restated): [N,C,H,W] -> [3,H',W'] with `nrow`
    images per row, `padding` pixels of `pad_value` around every image, single-channel images repeated to 3 channels."""
    if images.dim() == 3:
        images = images.unsqueeze(0)
    if images.size(1) == 1:
        images = images.expand(-1, 3, -1, -1)
    n, c, h, w = images.shape
    xmaps = min(nrow, n)
    ymaps = -(-n // xmaps)
    hh, ww = h + padding, w + padding
    grid = images.new_full((c, hh * ymaps + padding, ww * xmaps + padding), pad_value)
    for k in range(n):
        y, x = divmod(k, xmaps)
        grid[:, y * hh + padding:y * hh + padding + h, x * ww + padding:x * ww + padding + w] = images[k]
    return grid


def save_image(images, path, nrow=8, padding=2):
    """torchvision.utils.save_image (train.py:305-306): the grid, x255 + 0.5, clamped, as an 8-bit RGB PNG."""
    from PIL import Image
    grid = make_grid(images.detach().float().cpu(), nrow=nrow, padding=padding)
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(path, format="PNG")


def save_model(epoch, model, optimizer, loss, path):
    """Same on-disk dict as util.py:16-22 (epoch / model_state_dict / optimizer_state_dict / loss)."""
    state = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    torch.save({"epoch": epoch, "model_state_dict": state, "optimizer_state_dict": optimizer.state_dict(), "loss": loss}, path)


def load_model(path, model, device, optimizer=None):
    """util.py:36-42.  The dict holds tensors, ints and the optimizer state_dict only, so the no-code loader suffices:
    a reference-trained or third-party checkpoint is never unpickled with arbitrary-code execution."""
    ckpt = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return ckpt["epoch"]


def freeze(model):
    for p in model.parameters():
        p.requires_grad_(False)


def unfreeze(model):
    for p in model.parameters():
        p.requires_grad_(True)


def zero_grad(model):
    for p in model.parameters():
        p.grad = None
