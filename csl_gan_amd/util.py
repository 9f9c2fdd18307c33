"""Small helpers with the reference's names (util.py:10-42).  The reference's two dead functions
(max_batch_size_for_sample_rate, convert_modules: util.py:44-60 use undefined names) are not carried."""
import torch


def add_slash(path):
    if path is None:
        return None
    return path if path.endswith("/") else path + "/"


def denorm_celeba(img):
    return ((img + 1) / 2).clamp(0, 1)


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
