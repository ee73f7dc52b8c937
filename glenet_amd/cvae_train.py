"""The CVAE label-uncertainty generator's training step (BASELINE configs[3]) as one recorded launch sequence.

Our counterpart of cvae_uncertainty/train_utils/train_utils.py:50-72 -- zero_grad, Generator.forward (training
branch), `loss = reg_loss_post + anneal * lattent_loss + regular_loss`, backward, clip_grad_norm_(10), adam_onecycle
step (cfgs/exp20.yaml: lr 0.003, weight decay 0.01, betas (0.9, 0.99)) -- on glenet_amd.dense_path.CVAE:
PointNet extractors as row GEMMs (hipBLASLt, MFMA) + fused training BatchNorm (csrc/glx_bn.hip) up to
PointFeat.ROWS_MAX point rows (the reference's (B, C, P) modules beyond), the update as glx_adamw_clip_step on flat
buffers.  Shapes are static (B objects x P points), nothing reads back, so the step is
captured into ONE HIP graph; learning rate, beta1 and the annealing factor of the latent term are device scalars."""
import torch

from . import _lib
from .backbone import no_gc
from .optim import FlatAdamW

OPTIM_CFG = dict(LR=0.003, WEIGHT_DECAY=0.01, BETAS=(0.9, 0.99), GRAD_NORM_CLIP=10.0)     # cvae_uncertainty/cfgs/exp20.yaml


class CVAETrainStep:
    # the weight regulariser (sum of the parameter tensors' 2-norms) and its gradient from the flat buffers: 3 launches instead of
    # ~330 (four elementwise launches and an accumulation per parameter tensor behind autograd's norm backward)
    FLAT_REGULARISER = True

    def __init__(self, model, batch, num_points, lr=None, grad_clip=OPTIM_CFG["GRAD_NORM_CLIP"], device=None):
        dev = device if device is not None else next(model.parameters()).device
        self.model = model.train()
        c, l = model.x_encoder.fe.conv1.in_channels, model.latent_dim
        self.points = torch.zeros((batch, c, num_points), dtype=torch.float32, device=dev)
        self.cond = torch.zeros((batch, 8), dtype=torch.float32, device=dev)
        self.labels = torch.zeros((batch, 7), dtype=torch.float32, device=dev)
        self.eps = torch.zeros((batch, l), dtype=torch.float32, device=dev)
        self.anneal = torch.ones((), dtype=torch.float32, device=dev)          # linear_annealing(0, 1, epoch, total)
        self.draw_eps = True
        self.optimizer = FlatAdamW([p for p in model.parameters() if p.requires_grad],
                                   lr=lr if lr is not None else OPTIM_CFG["LR"], betas=OPTIM_CFG["BETAS"],
                                   weight_decay=OPTIM_CFG["WEIGHT_DECAY"], max_norm=grad_clip)
        self.graph = None
        self.loss = self.parts = self.terms = None
        self._regulariser_pending = False       # a flat regulariser value whose gradient has not been added yet
        self.optimizer.after_pack.append(self._add_regulariser_grad)

    def _add_regulariser_grad(self):
        if self._regulariser_pending:           # (only behind a step that took the regulariser from the flat buffer)
            self.optimizer.add_l2_norm_grad()
            self._regulariser_pending = False

    def load(self, points, gt_boxes_input, gt_boxes, eps=None):
        """Copy one batch into the step's static inputs (eps: the posterior's noise; None = drawn inside the step)."""
        self.points.copy_(points, non_blocking=True)
        self.cond.copy_(gt_boxes_input, non_blocking=True)
        self.labels.copy_(gt_boxes[:, :7], non_blocking=True)
        self.draw_eps = eps is None
        if eps is not None:
            self.eps.copy_(eps, non_blocking=True)

    def set_lr(self, lr, beta1=None, anneal=None):
        self.optimizer.set_lr(lr, beta1)
        if anneal is not None:
            self.anneal.fill_(float(anneal))

    def enqueue(self):
        self.loss = self.parts = self.terms = None
        self.model.zero_grad(set_to_none=True)
        if self.draw_eps:
            self.eps.normal_()
        flat = self.optimizer if self.FLAT_REGULARISER else None
        (reg, lat, regular), parts = self.model.training_losses(self.points, self.cond, self.labels, eps_post=self.eps,
                                                                flat_optimizer=flat)
        loss = reg + lat * self.anneal + regular
        if flat is None:
            loss.backward()
        else:              # the regulariser has no graph: its gradient goes into the flat buffer behind the others (optimizer.after_pack)
            (reg + lat * self.anneal).backward()
            self._regulariser_pending = True
        self.optimizer.step()
        self.loss, self.terms, self.parts = loss.detach(), (reg.detach(), lat.detach(), regular.detach()), parts
        return self.loss

    def capture(self, warmup=2):
        """Record the step.  Warm-up passes are real steps on the loaded batch; parameters, moments, step count and
        BatchNorm statistics are restored afterwards (as glenet_vr.StaticTrainStep.capture does)."""
        import gc
        gc.collect()                 # see backbone.StaticFramePipeline.capture: dead graphs pin AccumulateGrad streams
        opt = self.optimizer
        state = [opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.step_count, opt.hyper] + list(self.model.buffers())
        snap = [t.detach().clone() for t in state]
        dev = self.points.device
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.enqueue()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = _lib.new_graph()
        with torch.cuda.graph(self.graph, stream=side), no_gc():
            self.enqueue()
        self.memsets_replaced = _lib.finish_graph(self.graph)       # ROCm 7.2: memset nodes replay a stale pattern
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for t, s in zip(state, snap):
                t.copy_(s)
        self._written = list(self.model.parameters()) + list(self.model.buffers())
        _lib.bump_weights_epoch(self._written)
        return self

    def step(self):
        if self.graph is None:
            return self.enqueue()
        self.graph.replay()
        _lib.bump_weights_epoch(self._written)       # parameters and running statistics of THIS model moved
        return self.loss

    @staticmethod
    def flops_per_object(num_points, widths=(64, 128, 512), cin=4, latent=8):
        """Multiply-add flops (x2) of one object's forward: two large extractors + the narrow decoder extractor; the
        fully connected layers are noise next to them.  A training step is ~3x (forward, input and weight gradients)."""
        big = 2 * num_points * (cin * widths[0] + widths[0] * widths[1] + widths[1] * widths[2])
        small = 2 * num_points * (cin * 8 + 8 * 8 + 8 * 8)
        return 2 * big + small
