"""GShellFlexiCubesGeometry with the reference's class surface (geometry/gshell_flexicubes_geometry.py:45-364): the
FlexiCubes twin of GShellTetsGeometry -- voxel grid instead of tets, per-cube weights [n_cubes, 21] (registered under the
name `weight` as well, reference :97), the L_dev regulariser (x 0.25, :358) and deformation bound = mean edge length / 4."""
import torch

from ..render import mesh, optixutils as ou, render, util
from .gshell_flexicubes import GShellFlexiCubes
from .gshell_tets_geometry import GShellTetsGeometry, compute_sdf_reg_loss, sample_points   # noqa: F401  (same losses)
from .mlp import MLP, EdgeList, check_forward_status, forward_row_sharded, forward_row_sparse_backward


class GShellFlexiCubesGeometry(GShellTetsGeometry):
    def __init__(self, grid_res, scale, FLAGS):
        torch.nn.Module.__init__(self)
        self.FLAGS, self.grid_res, self.scale = FLAGS, grid_res, scale
        self.gflexicubes = GShellFlexiCubes()
        verts, indices = self.gflexicubes.construct_voxel_grid(grid_res)
        self.boxscale = torch.tensor(FLAGS.boxscale).view(1, 3).cuda()
        with torch.no_grad():
            self.optix_ctx = ou.OptiXContext()
        self.verts = verts * scale * self.boxscale
        self.indices = indices
        self.offset = 0.0
        self.generate_edges()
        n_cubes = indices.shape[0]
        if FLAGS.use_sdf_mlp:
            self.sdf = torch.nn.Parameter(torch.zeros_like(self.verts[:, 0]), requires_grad=True)
            self.sdf_net = MLP(skip_in=FLAGS.skip_in, n_freq=FLAGS.n_freq, n_hidden=FLAGS.n_hidden, d_hidden=FLAGS.d_hidden,
                               use_float16=FLAGS.use_float16).cuda()
            opt = torch.optim.Adam(self.sdf_net.parameters(), lr=1e-3)
            target = (self.verts / self.boxscale).norm(dim=1, keepdim=True) - FLAGS.sphere_init_norm
            for _ in range(FLAGS.sdf_mlp_pretrain_steps):
                loss = (self.sdf_net(self.verts) - target).pow(2).mean()
                opt.zero_grad()
                loss.backward()
                opt.step()
        else:
            sdf = torch.rand_like(self.verts[:, 0]) - 0.1 if not FLAGS.sphere_init else (self.verts / self.boxscale).norm(dim=1) - 0.5
            self.sdf = torch.nn.Parameter(sdf.clone().detach(), requires_grad=True)
        self.per_cube_weights = torch.nn.Parameter(torch.ones((n_cubes, 21), dtype=torch.float, device='cuda'), requires_grad=True)
        self.register_parameter('weight', self.per_cube_weights)
        msdf = (torch.rand_like(self.verts[:, 0]) - 0.01).clamp(-1, 1)
        self.msdf = torch.nn.Parameter(msdf.clone().detach(), requires_grad=True)
        self.deform = torch.nn.Parameter(torch.zeros_like(self.verts), requires_grad=True)
        self.clamp_deform()

    @torch.no_grad()
    def generate_edges(self):
        topo = self.gflexicubes.topology(self.indices, self.verts.shape[0], self.grid_res)
        e = topo.edges.long()
        self._all_edges = torch.sort(e, dim=1).values.int().contiguous()     # unique (min,max) grid edges (reference :124-127); read through the base class property `all_edges`
        self.max_displacement = util.length(self.verts[e[:, 0]] - self.verts[e[:, 1]]).mean() / 4

    def _refine_edges(self):
        # the extraction consumes SDF signs at every grid vertex and values only at the end points of sign-changing cube edges
        # (gshell_flexicubes.py:387-485: u_e of the surface edges), the sign regulariser runs over the same edges (:124-127): the
        # two-pass forward's selection rule over the unique cube edges
        el = getattr(self, "_edge_list", None)
        if el is None or el.edges.data_ptr() != self.all_edges.data_ptr():
            el = self._edge_list = EdgeList(self.all_edges, self.verts.shape[0])
        return el

    def getMesh(self, material, _training=False):
        v_deformed = self.verts + self.max_displacement * self.deform
        sdf = self._sdf_values(v_deformed)
        w = self.per_cube_weights
        out = self.gflexicubes(v_deformed, sdf, self.msdf, self.indices, self.grid_res, w[:, :12], w[:, 12:20], w[:, 20], training=_training)
        if self.FLAGS.use_sdf_mlp:
            # status words of the (two-pass) forward; the extraction has just synchronised the stream for its counts
            todo = check_forward_status(self.sdf_net)
            if todo is not None:
                self._recover_forward(todo)
                sdf = self._sdf_values(v_deformed)
                out = self.gflexicubes(v_deformed, sdf, self.msdf, self.indices, self.grid_res, w[:, :12], w[:, 12:20], w[:, 20], training=_training)
                check_forward_status(self.sdf_net)
        if len(out) == 3:
            raise RuntimeError("FlexiCubes produced an empty surface (the reference fails here too: its 3-tuple cannot be unpacked, :179)")
        verts, faces, reg_loss, extra = out
        self.gflexi_reg_loss = reg_loss.mean()
        imesh = mesh.Mesh(verts, faces, material=material)
        with torch.no_grad():
            ou.optix_build_bvh(self.optix_ctx, imesh.v_pos.contiguous(), imesh.faces_i32(), rebuild=1)
        imesh = mesh.auto_normals(imesh)
        self.last_mesh_sizes = (int(imesh.v_pos.shape[0]), int(imesh.t_pos_idx.shape[0]))
        d = {'imesh': imesh, 'sdf': sdf, 'msdf': extra['msdf'], 'msdf_watertight': extra['msdf_watertight'],
             'msdf_boundary': extra['msdf_boundary'], 'n_verts_watertight': extra['n_verts_watertight']}
        if getattr(self.FLAGS, "visualize_watertight", False):
            d['imesh_watertight'] = mesh.auto_normals(mesh.Mesh(extra['vertices_watertight'], extra['faces_watertight'], material=material))
        return d

    def tick(self, glctx, target, lgt, opt_material, loss_fn, iteration, denoiser):
        img_loss, depth_loss, reg_loss = GShellTetsGeometry.tick(self, glctx, target, lgt, opt_material, loss_fn, iteration, denoiser)
        flexi_reg = self.gflexi_reg_loss * 0.25                                # reference :358
        self.last_terms.add('global', flexi_reg)
        return img_loss, depth_loss, reg_loss + flexi_reg
