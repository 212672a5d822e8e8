"""Binary PLY point-cloud I/O (host side, numpy only).

Same two entry points and return conventions as the reference's utils/ply.py (read_ply :116-196, write_ply :217-328),
which the kernel-point cache (kernels/kernel_points.py:190-280) and the demo fragments (demo_data/*.ply) go through:

    read_ply(filename)                          -> structured array, one named field per vertex property
    write_ply(filename, field_list, field_names) -> True / False

Supported: `binary_little_endian` / `binary_big_endian` files whose first element is `vertex` with scalar
properties (what CloudCompare and the reference itself write).  ASCII files are rejected like the reference does (:131).
"""
import numpy as np

_PLY_TO_NP = {
    'char': 'i1', 'int8': 'i1', 'uchar': 'u1', 'uint8': 'u1', 'b1': 'u1',
    'short': 'i2', 'int16': 'i2', 'ushort': 'u2', 'uint16': 'u2',
    'int': 'i4', 'int32': 'i4', 'uint': 'u4', 'uint32': 'u4',
    'float': 'f4', 'float32': 'f4', 'double': 'f8', 'float64': 'f8',
}
_NP_TO_PLY = {'i1': 'char', 'u1': 'uchar', 'i2': 'short', 'u2': 'ushort', 'i4': 'int', 'u4': 'uint', 'f4': 'float',
              'f8': 'double'}
_ENDIAN = {'binary_little_endian': '<', 'binary_big_endian': '>'}


def _parse_header(f):
    if b'ply' not in f.readline():
        raise ValueError('The file does not start whith the word ply')
    fmt = f.readline().split()[1].decode()
    if fmt == 'ascii':
        raise ValueError('The file is not binary')
    if fmt not in _ENDIAN:
        raise ValueError('Unknown PLY format : ' + fmt)
    ext = _ENDIAN[fmt]
    elements = []        # [name, count, [(prop, dtype) ...]]
    while True:
        line = f.readline()
        if not line:
            raise ValueError('PLY header is not terminated')
        tok = line.split()
        if not tok:
            continue
        if tok[0] == b'end_header':
            break
        if tok[0] == b'element':
            elements.append([tok[1].decode(), int(tok[2]), []])
        elif tok[0] == b'property' and elements:
            if tok[1] == b'list':
                elements[-1][2].append(('list', (tok[2].decode(), tok[3].decode(), tok[4].decode())))
            else:
                elements[-1][2].append((tok[2].decode(), ext + _PLY_TO_NP[tok[1].decode()]))
    return ext, elements


def read_ply(filename, triangular_mesh=False):
    """-> structured numpy array of the vertex element (or [vertices, faces int32[F,3]] with triangular_mesh)."""
    with open(filename, 'rb') as f:
        ext, elements = _parse_header(f)
        if not elements or elements[0][0] != 'vertex':
            raise ValueError('first PLY element must be "vertex"')
        _, n, props = elements[0]
        if any(p[0] == 'list' for p in props):
            raise ValueError('list properties on vertices are not supported')
        data = np.fromfile(f, dtype=props, count=n)
        if data.shape[0] != n:
            raise ValueError('PLY file truncated: %d of %d vertices' % (data.shape[0], n))
        if not triangular_mesh:
            return data
        faces = np.zeros((0, 3), np.int32)
        if len(elements) > 1 and elements[1][0] == 'face':
            fd = np.fromfile(f, dtype=[('k', ext + 'u1'), ('v1', ext + 'i4'), ('v2', ext + 'i4'), ('v3', ext + 'i4')],
                             count=elements[1][1])
            faces = np.vstack((fd['v1'], fd['v2'], fd['v3'])).T
        return [data, faces]


def ply_vertex_count(filename):
    """Number of vertices, from the header only (the sharded runner balances fragments by it without reading them)."""
    with open(filename, 'rb') as f:
        _, elements = _parse_header(f)
    if not elements or elements[0][0] != 'vertex':
        raise ValueError('first PLY element must be "vertex"')
    return int(elements[0][1])


def read_ply_records(filename):
    """The vertex element as RAW bytes plus its record layout, for decoding on the GPU (ops.decode_xyz_records):
    -> (uint8 array [n * stride], dict(n, stride, offsets=(ox, oy, oz), dtype='f4'|'f8', big_endian=bool))."""
    with open(filename, 'rb') as f:
        ext, elements = _parse_header(f)
        if not elements or elements[0][0] != 'vertex':
            raise ValueError('first PLY element must be "vertex"')
        _, n, props = elements[0]
        if any(p[0] == 'list' for p in props):
            raise ValueError('list properties on vertices are not supported')
        dt = np.dtype(props)
        offs, kinds = [], set()
        for ax in ('x', 'y', 'z'):
            if ax not in dt.fields:
                raise ValueError('PLY vertex element has no property "%s"' % ax)
            fdt, off = dt.fields[ax][0], dt.fields[ax][1]
            offs.append(int(off))
            kinds.add(fdt.str[1:])
        if len(kinds) != 1 or next(iter(kinds)) not in ('f4', 'f8'):
            raise ValueError('x / y / z must share one floating type (float or double)')
        raw = np.fromfile(f, dtype=np.uint8, count=n * dt.itemsize)
        if raw.shape[0] != n * dt.itemsize:
            raise ValueError('PLY file truncated')
    return raw, dict(n=int(n), stride=int(dt.itemsize), offsets=tuple(offs), dtype=next(iter(kinds)), big_endian=(ext == '>'))


def read_ply_xyz(filename):
    """float32 [N,3] of the x/y/z properties (contiguous)."""
    d = read_ply(filename)
    return np.ascontiguousarray(np.stack([d['x'], d['y'], d['z']], axis=1).astype(np.float32))


def write_ply(filename, field_list, field_names, triangular_faces=None):
    """field_list: array or list of arrays ([N] or [N,k]); field_names: one name per column.  Little endian."""
    field_list = list(field_list) if isinstance(field_list, (list, tuple)) else [field_list]
    cols = []
    for i, fld in enumerate(field_list):
        fld = np.asarray(fld)
        if fld.ndim < 2:
            fld = fld.reshape(-1, 1)
        if fld.ndim > 2:
            print('fields have more than 2 dimensions')
            return False
        cols.append(fld)
    n = [c.shape[0] for c in cols]
    if not np.all(np.equal(n, n[0])):
        print('wrong field dimensions')
        return False
    if sum(c.shape[1] for c in cols) != len(field_names):
        print('wrong number of field names')
        return False
    if not filename.endswith('.ply'):
        filename += '.ply'
    dtype, k = [], 0
    for c in cols:
        code = c.dtype.str[1:]
        if code not in _NP_TO_PLY:
            print('unsupported dtype', c.dtype)
            return False
        for _ in range(c.shape[1]):
            dtype.append((field_names[k], '<' + code))
            k += 1
    rec = np.empty(n[0], dtype=dtype)
    k = 0
    for c in cols:
        for j in range(c.shape[1]):
            rec[field_names[k]] = c[:, j]
            k += 1
    with open(filename, 'wb') as f:
        lines = ['ply', 'format binary_little_endian 1.0', 'element vertex %d' % n[0]]
        lines += ['property %s %s' % (_NP_TO_PLY[t[1:]], name) for name, t in dtype]
        if triangular_faces is not None:
            lines += ['element face %d' % len(triangular_faces), 'property list uchar int vertex_indices']
        lines.append('end_header')
        f.write(('\n'.join(lines) + '\n').encode())
        rec.tofile(f)
        if triangular_faces is not None:
            tf = np.asarray(triangular_faces, dtype=np.int32)
            fr = np.empty(tf.shape[0], dtype=[('k', 'u1'), ('0', '<i4'), ('1', '<i4'), ('2', '<i4')])
            fr['k'] = 3
            fr['0'], fr['1'], fr['2'] = tf[:, 0], tf[:, 1], tf[:, 2]
            fr.tofile(f)
    return True
