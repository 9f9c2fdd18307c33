import sys, os, torch
sys.path.insert(0, "/root/repo")
from csl_gan_amd import ops
torch.manual_seed(0)
dev = "cuda:0"
def check(name, fn, inputs_list):
    """fn(*static_inputs) -> output; capture once, replay with each input set, compare with eager."""
    static = [t.clone() for t in inputs_list[0]]
    for _ in range(2):
        fn(*static)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn(*static)
    worst = 0.0
    for k, ins in enumerate(inputs_list):
        for s, t in zip(static, ins):
            s.copy_(t)
        g.replay()
        torch.cuda.synchronize()
        got = out.clone()
        want = fn(*[t.clone() for t in ins])
        torch.cuda.synchronize()
        err = ((got - want).abs().max() / (want.abs().max() + 1e-30)).item()
        worst = max(worst, err)
        print("  %s replay %d rel err %.3e finite %s" % (name, k, err, bool(torch.isfinite(got).all())))
    return worst
for N in (16, 32):
    mk = lambda *shape: [torch.randn(*shape, device=dev) for _ in range(3)]
    # conv4 dense wgrad: gy [N,4,4,512], x [N,8,8,256]
    gys, xs = mk(N, 4, 4, 512), mk(N, 8, 8, 256)
    check("wgrad_dense conv4 N=%d" % N, lambda gy, x: ops.conv2d_wgrad_dense(gy, x, 5, 5, stride=2, pad=2), list(zip(gys, xs)))
    gys, xs = mk(N, 8, 8, 256), mk(N, 16, 16, 128)
    check("wgrad_dense conv3 N=%d" % N, lambda gy, x: ops.conv2d_wgrad_dense(gy, x, 5, 5, stride=2, pad=2), list(zip(gys, xs)))
    gys, xs = mk(N, 16, 16, 128), mk(N, 32, 32, 64)
    check("wgrad_dense conv2 N=%d" % N, lambda gy, x: ops.conv2d_wgrad_dense(gy, x, 5, 5, stride=2, pad=2), list(zip(gys, xs)))
    gys, xs = mk(N, 32, 32, 64), mk(N, 64, 64, 3)
    check("wgrad_dense conv1 N=%d" % N, lambda gy, x: ops.conv2d_wgrad_dense(gy, x, 5, 5, stride=2, pad=2), list(zip(gys, xs)))
    ts = mk(N, 12288)
    check("row_l2norm N=%d" % N, lambda t: ops.row_l2norm(t), [(t,) for t in ts])
    # dgrad with a non-parameter filter (the sweeps' Dgrad(gy, ggw))
    gys, ws = mk(N, 4, 4, 512), mk(512, 5, 5, 256)
    check("dgrad conv4 N=%d" % N, lambda gy, w: ops.conv2d_dgrad(gy, w, (8, 8), stride=2, pad=2), list(zip(gys, ws)))
    gys, ws = mk(N, 16, 16, 128), mk(128, 5, 5, 64)
    check("dgrad conv2 N=%d" % N, lambda gy, w: ops.conv2d_dgrad(gy, w, (32, 32), stride=2, pad=2), list(zip(gys, ws)))
    xs, ws = mk(N, 8, 8, 256), mk(512, 5, 5, 256)
    check("fwd conv4 N=%d" % N, lambda x, w: ops.conv2d_fwd(x, w, None, stride=2, pad=2), list(zip(xs, ws)))
    xs, ws = mk(N, 32, 32, 64), mk(128, 5, 5, 64)
    check("fwd conv2 N=%d" % N, lambda x, w: ops.conv2d_fwd(x, w, None, stride=2, pad=2), list(zip(xs, ws)))
