"""Optimizer selection by config, mirroring builders/optimizer_builder.py:24-122 of the reference.

The reference returns a tf.train optimizer (Adam for model 000, wrapped in MovingAverageOptimizer) fed by
tf.train.exponential_decay.  Here `build` returns an object that owns the same three things over the trainable
net's FLAT fp32 parameter buffer: the learning-rate schedule, the fused Adam step (mpsr_adam_step) and the
moving average of the parameters.  (The reference saves with a plain tf.train.Saver, core/trainer.py:85: a
checkpoint holds the RAW variables under their own names and the averages under `<name>/ExponentialMovingAverage`.)
"""
import math


def _create_learning_rate(optimizer_config):
    """optimizer_builder.py:85-122 -> a function global_step -> learning rate."""
    learning_rate_type = optimizer_config.learning_rate_type
    if learning_rate_type == 'constant_learning_rate':
        lr = float(optimizer_config.learning_rate)
        return lambda global_step: lr
    if learning_rate_type == 'exponential_decay':
        lr0 = float(optimizer_config.initial_learning_rate)
        decay_steps = float(optimizer_config.decay_steps)
        decay_factor = float(optimizer_config.decay_factor)
        staircase = bool(optimizer_config.staircase)

        def schedule(global_step):  # tf.train.exponential_decay
            p = global_step / decay_steps
            return lr0 * decay_factor ** (math.floor(p) if staircase else p)
        return schedule
    raise ValueError('Learning rate {} not supported.'.format(learning_rate_type))


class AdamWithMovingAverage:
    """tf.train.AdamOptimizer(lr(global_step)) [+ tf.contrib.opt.MovingAverageOptimizer(average_decay)]."""

    def __init__(self, learning_rate, use_moving_average, moving_average_decay):
        self.learning_rate = learning_rate
        self.use_moving_average = use_moving_average
        self.moving_average_decay = moving_average_decay
        self.shadow = None

    def apply_gradients(self, net, global_step):
        """One update of net.params from net.grads at `global_step` (0-based, as TF's global_step before the
        increment); returns the learning rate used."""
        lr = self.learning_rate(global_step)
        net.adam_step(lr=lr)
        if self.use_moving_average:
            if self.shadow is None:
                # ExponentialMovingAverage initialises each shadow variable to the variable's initial value; the
                # first update then moves it -- starting from the post-step value differs by (1-decay) * one step
                self.shadow = net.params.clone()
            else:
                self.shadow.lerp_(net.params, 1.0 - self.moving_average_decay)
        return lr

    def apply_clipped_gradients(self, net, global_step, clip_table, clip_norm):
        """apply_gradients with slim's per-variable clip_by_norm (core/trainer.py:78-81) in front, fused: squared norms,
        then ONE pass over the flat buffers for clip -> Adam -> moving average (TrainNet.clip_adam_ema_step).  The step
        that creates the moving average takes it from the post-step parameters, as apply_gradients does."""
        lr = self.learning_rate(global_step)
        if self.use_moving_average and self.shadow is not None:
            net.clip_adam_ema_step(clip_table, clip_norm, lr=lr, shadow=self.shadow, ema_decay=self.moving_average_decay)
        else:
            net.clip_adam_ema_step(clip_table, clip_norm, lr=lr)
            if self.use_moving_average:
                self.shadow = net.params.clone()
        return lr

    def averaged_params(self, net):
        """The moving averages -- what a checkpoint holds under `<name>/ExponentialMovingAverage` -- or the raw
        parameters when averaging is off."""
        return self.shadow if self.shadow is not None else net.params


def build(optimizer_config, global_summaries=None, global_step=None):
    optimizer_type = optimizer_config.optimizer_type
    if optimizer_type != 'adam_optimizer':
        # rms_prop / momentum / gradient_descent are selectable in the reference but unused by any MonoPSR config
        raise ValueError('Optimizer %s not supported.' % optimizer_type)
    cfg = optimizer_config.adam_optimizer
    return AdamWithMovingAverage(_create_learning_rate(cfg), bool(cfg.use_moving_average),
                                 float(getattr(cfg, 'moving_average_decay', 0.9999)))
