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
    def __init__(self, vertical_scaling=0, x_min=0., width=0., vertical_offset=0., min_loss=0.005,
                 std_dev_multiplier=0.7, std_dev_shift=2):
        super().__init__()
        self.sigmas = torch.tensor([], dtype=torch.float32)
        self.losses = torch.tensor([], dtype=torch.float32)
        self.positions = torch.tensor([], dtype=torch.int64)
        self.history_size = 10000
        self.fourier_approximator = FourierSeriesFit(-torch.pi, torch.pi, num_terms=4)
        self.min_loss = min_loss

    @torch.no_grad()
    def add_data(self, sigmas, losses):
        if _dist_on() and dist.get_rank() != 0:
            return
        positions = torch.arange(sigmas.numel()) % sigmas.shape[1]
        h = self.history_size
        self.sigmas = torch.cat((self.sigmas, sigmas.flatten().detach().float().cpu()))[-h:]
        self.losses = torch.cat((self.losses, losses.flatten().detach().float().cpu()))[-h:]
        self.positions = torch.cat((self.positions, positions))[-h:]

    @torch.no_grad()
    def calculate_mean_loss(self, sigma):
        return self.fourier_approximator(sigma)

    def fit_loss_curve(self, sigmas=None, losses=None):
        self.fourier_approximator.fit_data(self.sigmas if sigmas is None else sigmas,
                                           self.losses if losses is None else losses)
