"""MultiNoiseLoss / FourierSeriesFit: per-sigma loss normalisation (reference edm2/loss_weight.py:9-162).
History lives on the host (rank 0 only), the 7 Fourier coefficients are a non-trainable parameter
`fourier_approximator.coefficients` (state_dict compatible) broadcast from rank 0 after each fit."""
import torch
from torch import nn
import torch.distributed as dist


def _dist_on():
    return dist.is_available() and dist.is_initialized()


class FourierSeriesFit(nn.Module):
    def __init__(self, interval_min, interval_max, num_terms=8):
        super().__init__()
        self.interval_min, self.interval_max, self.num_terms = interval_min, interval_max, num_terms
        self.num_basis = 2 * num_terms - 1
        self.coefficients = nn.Parameter(torch.zeros(self.num_basis, 1), requires_grad=False)
        self.coefficients_history = []

    def fourier_series(self, x):
        xl = torch.log10(x)
        basis = [torch.full_like(xl, 0.5)]
        for n in range(1, self.num_terms):
            basis += [torch.cos(n * xl), torch.sin(n * xl)]
        return torch.stack(basis, dim=-1)

    @torch.no_grad()
    def fit_data(self, X, Y):
        rank = dist.get_rank() if _dist_on() else 0
        if rank == 0:
            xl = torch.log10(X)
            keep = (xl >= self.interval_min) & (xl <= self.interval_max)
            basis = self.fourier_series(X[keep].flatten())
            sol = torch.linalg.lstsq(basis, Y[keep].flatten().log10().unsqueeze(1)).solution
            self.coefficients.data.copy_(sol)
            self.coefficients_history.append(sol.detach().clone())
        if _dist_on():
            dist.broadcast(self.coefficients.data, src=0)

    def forward(self, x):
        basis = self.fourier_series(x.reshape(-1))
        return (10 ** (basis @ self.coefficients.to(basis.device))).reshape(x.shape)


class MultiNoiseLoss(nn.Module):
    """History of the last `history_size` (sigma, loss, frame position) triples + the Fourier fit of log10 loss over
    log10 sigma (reference loss_weight.py:9-84).  On a HIP device the history is three preallocated rings and an entry
    counter IN DEVICE MEMORY, appended to by the loss kernel itself (ops.loss_tail -> oniris_loss_tail) with no host round
    trip; `.sigmas / .losses / .positions` (what the fit and the reference's dashboard read) copy them out in
    chronological order -- the only synchronisation, once per fit (every 500 * accumulation steps, gym_train.py:115-116).
    CPU tensors (tests without a GPU) keep the reference's concatenate-and-truncate lists."""

    def __init__(self, vertical_scaling=0, x_min=0., width=0., vertical_offset=0., min_loss=0.005,
                 std_dev_multiplier=0.7, std_dev_shift=2):
        super().__init__()
        self._host = (torch.tensor([], dtype=torch.float32), torch.tensor([], dtype=torch.float32),
                      torch.tensor([], dtype=torch.int64))
        self._ring = None                      # (ring_sigma, ring_loss, ring_pos, count) on the training device
        self.history_size = 10000
        self.fourier_approximator = FourierSeriesFit(-torch.pi, torch.pi, num_terms=4)
        self.min_loss = min_loss

    def device_history(self, device):
        """The rings the loss kernel appends to (created on first use; rank 0 only, like add_data: loss_weight.py:33-34)."""
        if _dist_on() and dist.get_rank() != 0:
            return None
        r = self._ring
        if r is None or r[0].device != device or r[0].numel() != self.history_size:
            h = self.history_size
            r = self._ring = (torch.zeros(h, dtype=torch.float32, device=device), torch.zeros(h, dtype=torch.float32, device=device),
                              torch.zeros(h, dtype=torch.int32, device=device), torch.zeros(1, dtype=torch.int64, device=device))
        return r

    def _chronological(self):
        hs, hl, hp = self._host
        if self._ring is None:
            return hs, hl, hp
        rs, rl, rp, cnt = self._ring
        n, cap = int(cnt.item()), rs.numel()
        if n <= cap:
            order = torch.arange(n)
        else:
            order = (torch.arange(cap) + n) % cap
        order = order.to(rs.device)
        ds, dl, dp = (z.index_select(0, order).cpu() for z in (rs, rl, rp))
        h = self.history_size
        return (torch.cat((hs, ds))[-h:], torch.cat((hl, dl))[-h:], torch.cat((hp, dp.to(torch.int64)))[-h:])

    sigmas = property(lambda self: self._chronological()[0])
    losses = property(lambda self: self._chronological()[1])
    positions = property(lambda self: self._chronological()[2])

    @torch.no_grad()
    def add_data(self, sigmas, losses):
        if _dist_on() and dist.get_rank() != 0:
            return
        positions = torch.arange(sigmas.numel()) % sigmas.shape[1]
        if sigmas.is_cuda:                     # same append as the loss kernel's, through torch ops (eager-path callers)
            rs, rl, rp, cnt = self.device_history(sigmas.device)
            n, cap = sigmas.numel(), rs.numel()
            idx = (cnt + torch.arange(n, device=sigmas.device)) % cap
            rs.index_copy_(0, idx, sigmas.flatten().detach().float())
            rl.index_copy_(0, idx, losses.flatten().detach().float())
            rp.index_copy_(0, idx, positions.to(sigmas.device, torch.int32))
            cnt += n
            return
        h = self.history_size
        hs, hl, hp = self._host
        self._host = (torch.cat((hs, sigmas.flatten().detach().float().cpu()))[-h:],
                      torch.cat((hl, losses.flatten().detach().float().cpu()))[-h:], torch.cat((hp, positions))[-h:])

    @torch.no_grad()
    def calculate_mean_loss(self, sigma):
        return self.fourier_approximator(sigma)

    def fit_loss_curve(self, sigmas=None, losses=None):
        if sigmas is None or losses is None:
            hs, hl, _ = self._chronological()
            sigmas, losses = (hs if sigmas is None else sigmas), (hl if losses is None else losses)
        self.fourier_approximator.fit_data(sigmas, losses)
