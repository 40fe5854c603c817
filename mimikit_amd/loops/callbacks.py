"""``GenerateCallback`` (reference ``mimikit/loops/callbacks.py:155-169``): the hook a Lightning trainer calls at the end of
an epoch to run the generate loop.  Written against the two attributes it uses (``trainer.current_epoch``, the loop), so it
needs no pytorch_lightning import; registering it with a real trainer works because Lightning only looks up hook names."""

__all__ = ["GenerateCallback"]


class GenerateCallback:
    def __init__(self, generate_loop=None, every_n_epochs: int = 10):
        self.loop = generate_loop
        self.every_n_epochs = every_n_epochs

    def on_train_epoch_end(self, trainer, model=None):
        if (trainer.current_epoch + 1) % self.every_n_epochs != 0:
            return
        self.loop.template_vars = dict(epoch=trainer.current_epoch + 1)
        for _ in self.loop.run():      # the loop is a generator: draining it generates (and logs) every prompt batch
            continue
